#!/bin/bash
# one GPU visit: GPU parity tests, bench line, rocprofv3 kernel stats, PMC traffic passes.
# usage: scratch/gpu_round.sh TAG   (outputs under gpurun_out/TAG/)
set -o pipefail
TAG=${1:-r}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $OUT/gpu_tests.log 2>&1 || { tail -30 $OUT/gpu_tests.log; exit 1; }
tail -3 $OUT/gpu_tests.log
timeout -k 10 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
cat $OUT/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python bench.py --no-cpu-baseline > $OUT/prof.log 2>&1 || { tail -20 $OUT/prof.log; exit 1; }
cp $OUT/prof/*/*_kernel_stats.csv $OUT/kernel_stats.csv
head -8 $OUT/kernel_stats.csv
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1 || { tail -20 $OUT/pmc_fetch.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.log 2>&1 || { tail -20 $OUT/pmc_write.log; exit 1; }
python scratch/pmc_summary.py $OUT > $OUT/pmc_summary.txt && cat $OUT/pmc_summary.txt
