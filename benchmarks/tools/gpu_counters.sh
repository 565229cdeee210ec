#!/bin/bash
# rocprofv3 counter passes over the default bench workload (one pass per counter block: rocprofv3 does not
# split counters over passes).  usage: benchmarks/tools/gpu_counters.sh TAG  -> gpurun_out/TAG/r02_*.json
set -o pipefail
TAG=${1:-ctr}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="bench.py --steps 16 --warmup 0 --no-cpu-baseline --no-stage-profile"
timeout -s KILL 180 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $OUT/sq -- python $ARGS > $OUT/sq.log 2>&1 || { tail -20 $OUT/sq.log; exit 1; }
timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $ARGS > $OUT/pmc_fetch.log 2>&1 || { tail -20 $OUT/pmc_fetch.log; exit 1; }
timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $ARGS > $OUT/pmc_write.log 2>&1 || { tail -20 $OUT/pmc_write.log; exit 1; }
python benchmarks/tools/counters_summary.py $OUT 16777216 10 $OUT/r02
