#!/bin/bash
# one benchmarks/configs.py config with the default build and every benchmarks/tools/variants/*.so.  usage: gpu_cfg_sweep.sh CONFIG
CFG=${1:-2}
echo "default build:"; timeout -k 10 300 python benchmarks/configs.py $CFG 2>/dev/null | cut -c1-230
for so in benchmarks/tools/variants/*.so; do
  n=$(basename $so .so)
  echo "$n:"; GJX_LIB_PATH=$PWD/$so timeout -k 10 300 python benchmarks/configs.py $CFG 2>/dev/null | cut -c1-230 || exit 1
done
