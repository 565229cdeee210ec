#!/bin/bash
# GPU parity tests + one bench line.  usage: scratch/gpu_quick.sh TAG [bench args]
set -o pipefail
TAG=${1:-q}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $OUT/gpu_tests.log 2>&1 || { tail -40 $OUT/gpu_tests.log; exit 1; }
tail -2 $OUT/gpu_tests.log
timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
cat $OUT/bench.json
