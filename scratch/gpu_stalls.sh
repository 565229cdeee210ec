#!/bin/bash
# where do host stalls happen in later processes?  3 processes, 600 steps each, report steps > 2 ms
for i in 1 2 3 4; do
  if [ $i = 3 ]; then export SETTLE_SLEEP=1.0; fi
  if [ $i = 4 ]; then export SETTLE_SLEEP=0.1; fi
  timeout -k 10 200 python scratch/step_times.py 2097152 600 2>/dev/null | python -c "
import sys
for l in sys.stdin:
    if l.startswith('ms/step:'):
        v=[float(x) for x in l.split()[1:]]
        big=[(i,x) for i,x in enumerate(v) if x>2.0]
        import statistics
        print('proc $i: median %.3f  stalls(step,ms): %s' % (statistics.median(v), big[:20]))
" || exit 1
done
