#!/bin/bash
# bench line only (+ optional SQ counter pass).  usage: scratch/gpu_bench.sh TAG [bench args]
set -o pipefail
TAG=${1:-b}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
python -c "
import json,sys
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print('value %.4g  ms/step %.3f  kernel_ms %.3f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
