#!/bin/bash
# copy the summaries gpu_profile_refresh.sh left under gpurun_out/ into profiles/ (run here, after the GPU call)
R=${1:-r03}
cp gpurun_out/${R}_bench_stats/kernel_stats.csv profiles/${R}_bench_kernel_stats.csv
cp gpurun_out/${R}_bench/${R}_sq_counters.json gpurun_out/${R}_bench/${R}_pmc_traffic.json profiles/
cp gpurun_out/${R}_bench/${R}_kernel_resources.json profiles/${R}_bench_kernel_resources.json
cp gpurun_out/${R}_bench/${R}_kernel_stats.csv profiles/${R}_bench_steps16_kernel_stats.csv
for c in hmc outlier; do
  cp gpurun_out/${R}_$c/${R}_${c}_kernel_stats.csv gpurun_out/${R}_$c/${R}_${c}_sq_counters.json \
     gpurun_out/${R}_$c/${R}_${c}_pmc_traffic.json gpurun_out/${R}_$c/${R}_${c}_kernel_resources.json profiles/
done
cp gpurun_out/${R}_configs.jsonl profiles/${R}_configs.jsonl
