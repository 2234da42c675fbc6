#!/bin/bash
# PMC passes focused on the matrix-form cost volume (one frame each): issue mix, matrix pipe, memory path
#   gpurun -- 'bash tools/exp/prof_cvmm.sh TAG [cfg]'  -> gpurun_out/TAG/pmc_summary.txt (+ kernel_stats.md)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-cvmm}; CFG=${2:-c2}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; rm -rf /tmp/pmcmm_* /tmp/trace_mm
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
B="SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT"
E="TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
F="FETCH_SIZE TCC_HIT_sum TCP_TCC_READ_REQ_sum"
G="WRITE_SIZE TCC_MISS_sum TCC_EA0_RDREQ_sum"
i=0
for C_ in "$A" "$B" "$E" "$F" "$G"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $C_ -d /tmp/pmcmm_$i -- python $R/tools/prof_render.py 1 $CFG > $O/pmc_$i.log 2>&1
done
mkdir -p /tmp/pmcmm_all; mv /tmp/pmcmm_? /tmp/pmcmm_all/
python $R/tools/pmc_summary.py /tmp/pmcmm_all > $O/pmc_summary.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/trace_mm -o t -- python $R/tools/prof_render.py 3 $CFG > $O/trace.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/trace_mm -name '*.db' | head -1) 12 --last-frame 5 > $O/kernel_stats.md 2>&1
grep -A24 "cost_volume_mm" $O/pmc_summary.txt | head -40
head -20 $O/kernel_stats.md | cut -c1-150
