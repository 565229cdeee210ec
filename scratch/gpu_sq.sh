#!/bin/bash
# SQ counter pass over bench.py (instruction mix / stalls of the dominant kernel)
set -o pipefail
TAG=${1:-sq}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $OUT/sq -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/sq.log 2>&1 || { tail -20 $OUT/sq.log; exit 1; }
python - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/sq/*/*_counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        res[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in sorted(res.items()):
    if "k_loc" not in k: continue
    print(k, {c: sum(v)/len(v) for c, v in sorted(d.items())})
PY
