#!/bin/bash
# usage: build_variant.sh NAME "-DFOO=1 -DBAR=2"  -> matchnerf_amd/libmnerf_hip_NAME.so (select with MNERF_LIB)
set -e
cd "$(dirname "$0")/../../matchnerf_amd/csrc"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-use-amdgpu-trackers=1 $2"
hipcc $F -shared -o ../libmnerf_hip_$1.so -x hip api.cpp backward.hip composite.hip conv.hip cost_volume.hip decoder.hip encoder_block.hip geometry.hip instance_norm.hip qkv.hip render_chunk.hip window_attention.hip
echo built ../libmnerf_hip_$1.so
