#!/bin/bash
# bench every scratch/variants/*.so (kernel-variant tuning)
mkdir -p gpurun_out/variants
for so in scratch/variants/*.so; do
  n=$(basename $so .so)
  GJX_LIB_PATH=$PWD/$so timeout -k 10 200 python bench.py --no-cpu-baseline --steps 24 --warmup 10 > gpurun_out/variants/$n.json 2> gpurun_out/variants/$n.err || { echo "$n FAILED"; tail -5 gpurun_out/variants/$n.err; exit 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/variants/$n.json').read().strip().splitlines()[-1])
print('$n value %.4g  ms/step %.3f  kernel_ms %.3f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done
