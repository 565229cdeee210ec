#!/bin/bash
# SQ counters of one benchmarks/configs.py config.  usage: benchmarks/tools/gpu_ctr_cfg.sh TAG CONFIG KERNEL_SUBSTR
set -o pipefail
TAG=${1:-cc}; CFG=${2:-4}; KS=${3:-k_gol}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $OUT/sq1 -- python benchmarks/configs.py $CFG > $OUT/sq1.log 2>&1 || { tail -20 $OUT/sq1.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_ANY --output-format csv -d $OUT/sq2 -- python benchmarks/configs.py $CFG > $OUT/sq2.log 2>&1 || { tail -20 $OUT/sq2.log; exit 1; }
python - "$OUT" "$KS" <<'PY'
import csv, glob, sys, collections
out, ks = sys.argv[1], sys.argv[2]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/sq*/*/*_counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if ks in row["Kernel_Name"]:
            res[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in res.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"  {c:26s} {sum(v)/len(v):.4g}  (n={len(v)})")
PY
