#!/bin/bash
# benchmarks/tools/build_ubench.sh [extra hipcc -D flags for the library sources]: builds benchmarks/tools/ubench/hbm_stages
# against the library's objects (rebuilt into a scratch directory when flags are given)
set -e
cd "$(dirname "$0")/../.."
OBJDIR=genjax_amd/_lib/obj
if [ $# -gt 0 ]; then
  OBJDIR=/tmp/gjx_ubench_obj; mkdir -p $OBJDIR
  for f in genjax_amd/csrc/*.hip; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on "$@" -c $f -o $OBJDIR/$(basename $f).o &
  done
  wait
fi
OBJS=$(ls $OBJDIR/*.o | sed 's/^/-Wl,/' | tr '\n' ' ')
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude benchmarks/tools/ubench/hbm_stages.hip $OBJS -o benchmarks/tools/ubench/hbm_stages 2>&1 | grep -v warning | head -c 3000
