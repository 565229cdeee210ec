#!/bin/bash
# rocprofv3 kernel stats + PMC traffic passes of bench.py.  usage: scratch/gpu_prof.sh TAG [bench args]
set -o pipefail
TAG=${1:-p}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline "$@" > $OUT/prof.log 2>&1 || { tail -20 $OUT/prof.log; exit 1; }
cp $OUT/prof/*/*_kernel_stats.csv $OUT/kernel_stats.csv
head -12 $OUT/kernel_stats.csv
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/pmc_fetch.log 2>&1 || { tail -20 $OUT/pmc_fetch.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/pmc_write.log 2>&1 || { tail -20 $OUT/pmc_write.log; exit 1; }
python scratch/pmc_summary.py $OUT > $OUT/pmc_summary.txt && cat $OUT/pmc_summary.txt
