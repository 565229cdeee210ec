#!/bin/bash
# config 3 (HMC, K 25, L 10) with every benchmarks/tools/variants/hmc_w*.so.  usage: benchmarks/tools/gpu_hmc_sweep.sh [N]
mkdir -p gpurun_out/hmcsweep
export GJX_CFG3_N=${1:-4194304}
echo "default build:"; timeout -k 10 300 python benchmarks/configs.py 3 2>/dev/null | cut -c90-260
for so in benchmarks/tools/variants/hmc_w*.so; do
  n=$(basename $so .so)
  echo "$n:"; GJX_LIB_PATH=$PWD/$so timeout -k 10 300 python benchmarks/configs.py 3 2>/dev/null | cut -c90-260 || exit 1
done
