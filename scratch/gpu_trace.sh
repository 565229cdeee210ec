#!/bin/bash
# kernel trace + stats of one bench configuration.  usage: scratch/gpu_trace.sh TAG [bench args]
set -o pipefail
TAG=${1:-tr}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python bench.py --no-cpu-baseline "$@" > $OUT/prof.log 2>&1 || { tail -20 $OUT/prof.log; exit 1; }
grep -h '"metric"' $OUT/prof.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('value %.4g  ms/step %.3f  kernel_ms %.3f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
python - $OUT <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/prof/*/*_kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        print('%-60s %5s %12.1f %6s' % (r['Name'][:60], r['Calls'], float(r['AverageNs']), r['Percentage']))
PY
