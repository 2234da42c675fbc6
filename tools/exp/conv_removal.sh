#!/bin/bash
# Removal experiments on conv_kernel (CONV_EXP=n in conv.hip: wrong results, meaningful times): 1 no matrix instructions, 2 no operand
# loads, 3 no operand split, 4 no weight requests / segment barriers, 5 no tap geometry.
# Build here (CPU container), time on the GPU:  gpurun -- 'bash tools/exp/conv_removal.sh run'
cd "$(dirname "$0")/../.."
if [ "$1" = run ]; then
  python tools/exp/conv_time.py 2>&1 | grep -v amdgpu.ids
  for n in 1 2 3 4 5; do echo -n "exp $n: "; MNERF_LIB=$PWD/matchnerf_amd/libmnerf_hip_cvx$n.so python tools/exp/conv_time.py 2>&1 | grep -v amdgpu.ids; done
else
  for n in 1 2 3 4 5; do bash tools/exp/build_variant.sh cvx$n "-DCONV_EXP=$n -I$PWD/include" conv.hip & done; wait
fi
