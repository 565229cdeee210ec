#!/bin/bash
# rocprofv3 counter passes (one pass per counter block: rocprofv3 does not split counters over passes) and a
# kernel-stats pass over one workload.
# usage: benchmarks/tools/gpu_counters.sh TAG PREFIX PARTICLES MOVES [python-args ...]
#   -> gpurun_out/TAG/PREFIX_sq_counters.json, PREFIX_pmc_traffic.json, PREFIX_kernel_stats.csv
# default python-args: the default bench workload (bench.py --steps 16 --warmup 0 --no-cpu-baseline --no-stage-profile)
set -o pipefail
TAG=${1:-ctr}; PREFIX=${2:-r03}; PARTICLES=${3:-16777216}; MOVES=${4:-10}
shift; shift; shift; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ $# -eq 0 ]; then set -- bench.py --steps 16 --warmup 0 --no-cpu-baseline --no-stage-profile; fi
echo "python $*" > $OUT/command.txt
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $OUT/sq -- python "$@" > $OUT/sq.log 2>&1 || { tail -20 $OUT/sq.log; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python "$@" > $OUT/pmc_fetch.log 2>&1 || { tail -20 $OUT/pmc_fetch.log; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python "$@" > $OUT/pmc_write.log 2>&1 || { tail -20 $OUT/pmc_write.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python "$@" > $OUT/prof.log 2>&1 || { tail -20 $OUT/prof.log; exit 1; }
cp $(ls -t $OUT/prof/*/*_kernel_stats.csv | head -1) $OUT/${PREFIX}_kernel_stats.csv
python benchmarks/tools/counters_summary.py $OUT $PARTICLES $MOVES $OUT/$PREFIX "$(cat $OUT/command.txt)"
