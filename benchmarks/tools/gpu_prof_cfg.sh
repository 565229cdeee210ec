#!/bin/bash
# rocprofv3 kernel stats of one benchmarks/configs.py config.  usage: benchmarks/tools/gpu_prof_cfg.sh TAG CONFIG
set -o pipefail
TAG=${1:-pc}; CFG=${2:-1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python benchmarks/configs.py $CFG > $OUT/prof.log 2>&1 || { tail -20 $OUT/prof.log; exit 1; }
cp $OUT/prof/*/*_kernel_stats.csv $OUT/kernel_stats.csv
cp $OUT/prof/*/*_kernel_trace.csv $OUT/kernel_trace.csv
head -16 $OUT/kernel_stats.csv | cut -c1-150
grep '^{' $OUT/prof.log | tail -2
