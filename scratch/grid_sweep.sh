#!/bin/bash
# sweep of the persistent grid of the MH kernel (tuning experiment)
for g in 2048 2048 2560 4096 8192 16384 65536; do
  echo "grid $g" >> gpurun_out/grid_sweep.log
  GJX_HEAVY_GRID=$g timeout -k 10 120 python bench.py --steps 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_launch_ms'])" >> gpurun_out/grid_sweep.log || exit 1
done
