#!/bin/bash
# usage: build_timeline.sh [NAME [extra flags]]  -> matchnerf_amd/libmnerf_hip_NAME.so with the s_memtime stamps compiled in
exec "$(dirname "$0")/build_variant.sh" "${1:-tl}" "-DMNERF_TIMELINE $2" decoder.hip
