#!/bin/bash
# headline bench with every benchmarks/tools/variants/ext_w*.so (occupancy of k_loc_extend).  usage: gpu_ext_sweep.sh
mkdir -p gpurun_out/extsweep
show() { python -c "
import json,sys
l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
st={s['stage']:s['avg_ms'] for s in (l.get('roofline_stages') or [])}
print(sys.argv[2], 'ms/step %.4f' % l['ms_per_step'], 'extend %.4f' % st.get('extend', float('nan')), 'moves %.4f' % st.get('moves', float('nan')))" $1 $2; }
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 64 > gpurun_out/extsweep/default.json 2>/dev/null || exit 1
show gpurun_out/extsweep/default.json default
for so in benchmarks/tools/variants/ext_w*.so; do
  n=$(basename $so .so)
  GJX_LIB_PATH=$PWD/$so timeout -k 10 200 python bench.py --no-cpu-baseline --steps 64 > gpurun_out/extsweep/$n.json 2>/dev/null || exit 1
  show gpurun_out/extsweep/$n.json $n
done
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 64 > gpurun_out/extsweep/default2.json 2>/dev/null || exit 1
show gpurun_out/extsweep/default2.json default_again
