#!/bin/bash
# sweep of the grid cap of the per-particle kernels (tuning experiment)
rm -f gpurun_out/grid_sweep.log
for g in 8192 16384 32768 65536 16384; do
  GJX_HEAVY_GRID=$g timeout -k 10 120 python bench.py --steps 24 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($g, 'ms/step %.3f kernel %.3f' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))" >> gpurun_out/grid_sweep.log || exit 1
done
cat gpurun_out/grid_sweep.log
