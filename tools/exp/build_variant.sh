#!/bin/bash
# usage: build_variant.sh NAME "-DFOO=1 -DBAR=2" [sources...] -> matchnerf_amd/libmnerf_hip_NAME.so (select with MNERF_LIB)
# The listed sources (default: decoder.hip in both of its parts, cost_volume.hip) are rebuilt with the extra flags, every
# other object comes from the regular build (python -m matchnerf_amd.csrc.build first).
set -e
cd "$(dirname "$0")/../../matchnerf_amd/csrc"
NAME=$1; EXTRA=$2; shift; shift || true
SRCS=${@:-"decoder.hip cost_volume.hip"}
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-use-amdgpu-trackers=1 -fno-slp-vectorize -Xclang -target-feature -Xclang -packed-fp32-ops $EXTRA"
mkdir -p build/var
SKIP=""
NEW=""
for s in $SRCS; do
  b=${s%.*}
  hipcc $F -c $s -o build/var/${b}_$NAME.o &
  SKIP="$SKIP|$b.o"; NEW="$NEW build/var/${b}_$NAME.o"
  if [ "$b" = decoder ]; then
    hipcc $F -DMNERF_DECODER_PART=1 -c $s -o build/var/decoder_fused_$NAME.o &
    SKIP="$SKIP|decoder_fused.o"; NEW="$NEW build/var/decoder_fused_$NAME.o"
  fi
done
wait
OBJS=$(ls build/*.o | grep -v -E "/(${SKIP#|})$")
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmnerf_hip_$NAME.so $OBJS $NEW
echo built ../libmnerf_hip_$NAME.so
