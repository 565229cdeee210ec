#!/bin/bash
# The round's evidence in one go, each step under its own time limit: kernel stats + counter passes of the default
# bench, of the HMC config (N = 2^24, K 25, L 10) and of the outlier Gibbs+HMC config; then the default bench line.
# usage: benchmarks/tools/gpu_profile_refresh.sh ROUND   -> gpurun_out/ROUND_*/ (copy the summaries into profiles/)
R=${1:-r03}
bash benchmarks/tools/gpu_prof.sh ${R}_bench_stats || exit 1
bash benchmarks/tools/gpu_counters.sh ${R}_bench ${R} 16777216 10 || exit 1
bash benchmarks/tools/gpu_counters.sh ${R}_hmc ${R}_hmc 16777216 25 benchmarks/configs.py 3 || exit 1
bash benchmarks/tools/gpu_counters.sh ${R}_outlier ${R}_outlier 1048576 20 benchmarks/configs.py 2 || exit 1
timeout -k 10 400 python bench.py > gpurun_out/${R}_bench_line.json || exit 1
timeout -k 10 600 python benchmarks/configs.py > gpurun_out/${R}_configs.jsonl 2>/dev/null || exit 1
