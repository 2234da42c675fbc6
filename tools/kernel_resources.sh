#!/bin/bash
# usage: tools/kernel_resources.sh matchnerf_amd/csrc/decoder.hip [extra hipcc flags]
# One line per kernel: VGPRs, spilled VGPRs / SGPRs, scratch bytes per lane, occupancy.
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-use-amdgpu-trackers=1 -c "$src" -o /dev/null "$@" \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys
cur = None
def flush(c):
    if c: print("{fn:40s} VGPRs {VGPRs:>4s} AGPRs {AGPRs:>3s} vgpr-spill {VGPRs Spill:>4s} sgpr-spill {SGPRs Spill:>4s} scratch {ScratchSize:>5s} B/lane occ {Occupancy}".format(**c))
for line in sys.stdin:
    m = re.search(r"remark:\s+(Function Name|VGPRs Spill|SGPRs Spill|VGPRs|AGPRs|ScratchSize|Occupancy)[^:]*: (\S+)", line)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        flush(cur)
        v = re.sub(r"_Z14decoder_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E.*", r"decoder_kernel<\1,\2,\3,\4>", v)
        cur = {"fn": v[:40]}
    else:
        cur[k] = v
flush(cur)
'
