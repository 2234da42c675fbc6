#!/bin/bash
# PMC passes focused on the cost-volume kernel (one frame each).
cd /root/repo; export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/cv_pmc; mkdir -p $O
cd /tmp
rocprofv3 --list-avail > $O/avail.txt 2>&1
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
B="SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT"
C="TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum FETCH_SIZE"
D="TA_BUSY_avr TA_TA_BUSY_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum"
i=0
for C_ in "$A" "$B" "$C" "$D"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $C_ -d /tmp/pmc_$i -- python $R/tools/prof_render.py 1 > $O/pmc_$i.log 2>&1
done
python $R/tools/pmc_summary.py /tmp > $O/pmc_summary.txt 2>&1
