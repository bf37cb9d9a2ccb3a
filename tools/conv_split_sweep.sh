#!/bin/bash
# The end-of-grid piece count of the one-wave conv kernel (CMF_CONV_SPLIT_EXTRA, tiles per launch cut into pieces) at full
# size: is the default (3 tiles per CU = 768) still the best on this box?   bash tools/conv_split_sweep.sh [T=50000]
T=${1:-50000}
for extra in 0 256 512 768 1024 1536 2048; do
  echo "CMF_CONV_SPLIT_EXTRA=$extra"
  CMF_CONV_SPLIT_EXTRA=$extra python3 tools/time_kernels.py $T 20 | grep "conv_kernel=3"
done
