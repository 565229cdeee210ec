#!/bin/bash
# Rehearsal of the multi-rank bench flow on ONE GPU (gloo transport, ranks share the device).
# usage: benchmarks/tools/gpu_rehearse.sh TAG NRANKS [bench args]
set -o pipefail
TAG=${1:-reh}; NR=${2:-2}; shift; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $NR --master-addr 127.0.0.1 --master-port 29541 \
  bench.py --gpus $NR --backend gloo --no-cpu-baseline "$@" > $OUT/bench_g$NR.json 2> $OUT/bench_g$NR.err || { tail -30 $OUT/bench_g$NR.err; exit 1; }
tail -c 600 $OUT/bench_g$NR.json
