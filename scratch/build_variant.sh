#!/bin/bash
# build a tuning variant of the library: scratch/build_variant.sh NAME FILE(gjx_loc|gjx_models|...) [extra hipcc flags]
NAME=$1; FILE=$2; shift; shift
mkdir -p scratch/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=on -Iinclude -DGJX_DEV_R8_ONLY "$@" -c genjax_amd/csrc/$FILE.hip -o scratch/variants/$NAME.o || exit 1
OBJS=$(ls genjax_amd/_lib/obj/*.o | grep -v $FILE)
hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/variants/$NAME.so scratch/variants/$NAME.o $OBJS && rm scratch/variants/$NAME.o
