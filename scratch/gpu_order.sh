#!/bin/bash
# does a later process in the same call run slower?  prints cgroup cpu.stat and a CPU spin time around runs
set -o pipefail
OUT=gpurun_out/order; mkdir -p $OUT
stat() { cat /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; echo; python -c "
import time; t=time.perf_counter(); s=0
for i in range(3000000): s+=i
print('cpu spin %.3f s' % (time.perf_counter()-t))"; }
stat
for i in 1 2 3; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --particles 2097152 --steps 64 --warmup 8 > $OUT/b$i.json 2> $OUT/b$i.err || exit 1
  python -c "
import json; d=json.loads(open('$OUT/b$i.json').read().strip().splitlines()[-1]); print('run $i: ms/step %.3f kernel %.3f' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))"
  stat
  if [ $i = 2 ]; then sleep 20; fi
done
