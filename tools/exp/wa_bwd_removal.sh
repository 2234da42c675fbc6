#!/bin/bash
# Removal experiments on the split-fp16 attention backward (WBS_EXP=n in window_attention_backward.hip: wrong results, meaningful
# times): 1 no matrix instructions, 2 no role-2 image build, 3 no role-1 split + stores, 4 no softmax arithmetic, 5 no global fetch.
# Build here (CPU container), time on the GPU:  gpurun -- 'bash tools/exp/wa_bwd_removal.sh run'
cd "$(dirname "$0")/../.."
if [ "$1" = run ]; then
  python tools/exp/wa_bwd_alone.py 2>&1 | grep -v amdgpu.ids
  for n in 1 2 3 4 5; do MNERF_LIB=$PWD/matchnerf_amd/libmnerf_hip_wbx$n.so python tools/exp/wa_bwd_alone.py 2>&1 | grep -v amdgpu.ids; done
else
  for n in 1 2 3 4 5; do bash tools/exp/build_variant.sh wbx$n "-DWBS_EXP=$n -I$PWD/include" window_attention_backward.hip & done; wait
fi
