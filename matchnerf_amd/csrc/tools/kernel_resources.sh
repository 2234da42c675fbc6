#!/bin/bash
# usage: tools/kernel_resources.sh matchnerf_amd/csrc/decoder.hip [extra hipcc flags]
# One line per kernel: VGPRs, AGPRs, spilled VGPRs, scratch bytes, occupancy, LDS.
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$src" -o /dev/null "$@" \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys
cur = {}
for line in sys.stdin:
    m = re.search(r"remark: (?:.*?:\d+:\d+: )?\s*(Function Name|VGPRs|AGPRs|VGPR Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs): (\S+)", line)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        if cur: print(cur)
        cur = {"fn": v}
    else:
        cur[k.split(" [")[0]] = v
if cur: print(cur)
' | sed "s/_Z14decoder_kernelILi\([0-9]*\)ELi\([0-9]*\)ELi\([0-9]*\)E[^']*/decoder_kernel<\1,\2,\3>/"
