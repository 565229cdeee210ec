#!/bin/bash
# rocprofv3 kernel stats of the default bench command.  usage: benchmarks/tools/gpu_prof.sh TAG [bench args]
set -o pipefail
TAG=${1:-p}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python bench.py --no-cpu-baseline "$@" > $OUT/prof.log 2>&1 || { tail -20 $OUT/prof.log; exit 1; }
cp $OUT/prof/*/*_kernel_stats.csv $OUT/kernel_stats.csv
head -14 $OUT/kernel_stats.csv | cut -c1-200
