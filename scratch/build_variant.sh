#!/bin/bash
# build a tuning variant of the library: scratch/build_variant.sh NAME [extra hipcc flags for gjx_loc.hip]
NAME=$1; shift
mkdir -p scratch/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DGJX_DEV_R8_ONLY "$@" -c genjax_amd/csrc/gjx_loc.hip -o scratch/variants/$NAME.o || exit 1
OBJS=$(ls genjax_amd/_lib/obj/*.o | grep -v gjx_loc)
hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/variants/$NAME.so scratch/variants/$NAME.o $OBJS && rm scratch/variants/$NAME.o
