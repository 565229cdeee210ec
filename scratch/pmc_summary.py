"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes: HBM bytes per launch per kernel.

Corrections (MI355X_MICROARCH.md, HBM section): rocprofv3 reports both counters in KiB;
on gfx950 FETCH_SIZE tallies each 128-B read request as 64 B, so reads are doubled
(calibrated here on k_max_partial / k_norm_weights: a 64 MiB streamed read reports 32 790 KiB).
WRITE_SIZE is exact for streaming stores.  Writes <out>/pmc_traffic.json.
"""
import collections
import csv
import glob
import json
import sys

out = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for which in ("fetch", "write"):
    for f in glob.glob(f"{out}/pmc_{which}/*/*_counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")
            res[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("kernel,launches,FETCH_SIZE_KiB_avg_raw,WRITE_SIZE_KiB_avg_raw,hbm_read_bytes_corrected,hbm_write_bytes,hbm_bytes_per_launch")
js = {}
for k, d in sorted(res.items()):
    f = d.get("FETCH_SIZE", [])
    w = d.get("WRITE_SIZE", [])
    fa, wa = sum(f) / max(1, len(f)), sum(w) / max(1, len(w))
    rd, wr = 2.0 * fa * 1024.0, wa * 1024.0
    js[k] = {"launches": max(len(f), len(w)), "read_bytes": rd, "write_bytes": wr, "bytes": rd + wr}
    print(f"{k},{max(len(f), len(w))},{fa:.1f},{wa:.1f},{rd:.0f},{wr:.0f},{rd + wr:.0f}")
json.dump(js, open(f"{out}/pmc_traffic.json", "w"), indent=1, sort_keys=True)
