#!/bin/bash
exec "$(dirname "$0")/build_variant.sh" tl "-DMNERF_TIMELINE" decoder.hip
