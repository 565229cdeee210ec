#!/bin/bash
# build a tuning variant of the library: benchmarks/tools/build_variant.sh NAME FILE(gjx_loc|gjx_models|...) [extra hipcc flags]
NAME=$1; FILE=$2; shift; shift
mkdir -p benchmarks/tools/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=on -Iinclude -DGJX_DEV_R8_ONLY "$@" -c genjax_amd/csrc/$FILE.hip -o benchmarks/tools/variants/$NAME.o || exit 1
OBJS=$(ls genjax_amd/_lib/obj/*.o | grep -v $FILE)
hipcc --offload-arch=gfx950 -shared -fPIC -o benchmarks/tools/variants/$NAME.so benchmarks/tools/variants/$NAME.o $OBJS && rm benchmarks/tools/variants/$NAME.o
