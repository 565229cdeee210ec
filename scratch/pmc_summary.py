"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes: bytes per launch per kernel."""
import csv, glob, sys, collections
out = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for which in ("fetch", "write"):
    for f in glob.glob(f"{out}/pmc_{which}/*/*_counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0]
            res[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("kernel,launches,FETCH_SIZE_avg,WRITE_SIZE_avg (counter units as reported by rocprofv3)")
for k, d in sorted(res.items()):
    f = d.get("FETCH_SIZE", []); w = d.get("WRITE_SIZE", [])
    print(f"{k},{max(len(f), len(w))},{sum(f)/max(1,len(f)):.1f},{sum(w)/max(1,len(w)):.1f}")
