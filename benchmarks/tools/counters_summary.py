"""Summarise rocprofv3 --pmc passes over bench.py into the two files bench.py reads.

  python benchmarks/tools/counters_summary.py <dir with sq/ pmc_fetch/ pmc_write/> <particles> <moves> <out prefix> [command]

Writes <prefix>_sq_counters.json (SQ_* averages per launch per kernel) and <prefix>_pmc_traffic.json (HBM bytes
per launch per kernel).  Corrections (MI355X_MICROARCH.md, HBM section): rocprofv3 reports FETCH_SIZE and
WRITE_SIZE in KiB; on gfx950 FETCH_SIZE tallies each 128-B read request as 64 B, so reads are doubled
(calibrated on the streaming kernels of the same run: a 64 MiB read reports ~32 790 KiB); WRITE_SIZE is exact.
"""
import collections
import csv
import glob
import json
import sys

src, particles, moves, prefix = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
workload = sys.argv[5] if len(sys.argv) > 5 else ("python bench.py --steps 16 --warmup 0 --no-cpu-baseline "
                                                  "--no-stage-profile")


def collect(sub):
    res = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{src}/{sub}/*/*_counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")
            res[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return res


sq = collect("sq")
out = {"particles": particles, "moves": moves, "command": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM "
       "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES -- " + workload, "kernels": {}}
for k, d in sorted(sq.items()):
    e = {c: sum(v) / len(v) for c, v in sorted(d.items())}
    e.update({"launches": max(len(v) for v in d.values()), "particles": particles, "moves": moves})
    out["kernels"][k] = e
json.dump(out, open(f"{prefix}_sq_counters.json", "w"), indent=1, sort_keys=True)

fe, wr = collect("pmc_fetch"), collect("pmc_write")
out = {"particles": particles, "moves": moves, "unit": "bytes per launch (FETCH_SIZE KiB x 2 x 1024 + WRITE_SIZE KiB x 1024)",
       "command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (one pass each) -- " + workload, "kernels": {}}
for k in sorted(set(fe) | set(wr)):
    f, w = fe.get(k, {}).get("FETCH_SIZE", []), wr.get(k, {}).get("WRITE_SIZE", [])
    fa, wa = sum(f) / max(1, len(f)), sum(w) / max(1, len(w))
    rd, wb = 2.0 * fa * 1024.0, wa * 1024.0
    out["kernels"][k] = {"launches": max(len(f), len(w)), "read_bytes": rd, "write_bytes": wb, "bytes": rd + wb,
                         "FETCH_SIZE_KiB_raw": fa, "WRITE_SIZE_KiB_raw": wa, "particles": particles, "moves": moves}
json.dump(out, open(f"{prefix}_pmc_traffic.json", "w"), indent=1, sort_keys=True)
for k, v in out["kernels"].items():
    print(f"{k}: {v['bytes'] / particles:.1f} B/particle over {v['launches']} launches")

# launch resources per kernel from the kernel trace of the same workload (when the directory holds one)
res = {}
for f in glob.glob(f"{src}/prof/*/*_kernel_trace.csv"):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "")
        e = res.setdefault(name, {"launches": 0, "total_ns": 0.0})
        e["launches"] += 1
        e["total_ns"] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
        # (register counts are left out: the trace reports them in allocation granules, the compiler's remarks --
        #  hipcc -Rpass-analysis=kernel-resource-usage -- are the place to read them)
        for col, key_ in (("Scratch_Size", "scratch_bytes_per_lane"), ("LDS_Block_Size", "lds_bytes_per_block"),
                          ("Workgroup_Size_X", "workgroup"), ("Grid_Size_X", "grid_threads")):
            if col in row and row[col] != "":
                e[key_] = max(e.get(key_, 0), int(float(row[col])))
if res:
    for e in res.values():
        e["avg_us"] = e.pop("total_ns") / e["launches"] / 1e3
        if "scratch_bytes_per_lane" in e and "grid_threads" in e:
            e["scratch_bytes_per_launch_upper_bound"] = e["scratch_bytes_per_lane"] * e["grid_threads"]
    json.dump({"command": "rocprofv3 --kernel-trace --stats -- " + workload, "kernels": res},
              open(f"{prefix}_kernel_resources.json", "w"), indent=1, sort_keys=True)
