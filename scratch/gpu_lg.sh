#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/lg
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lg/prof -- python benchmarks/configs.py 1 > gpurun_out/lg/out.txt 2>&1 || { tail gpurun_out/lg/out.txt; exit 1; }
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/lg/prof/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    print("%-70s %6s %10.1f %6s"%(r["Name"][:70],r["Calls"],float(r["AverageNs"]),r["Percentage"]))
PY
