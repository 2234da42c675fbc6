#!/bin/bash
# PMC passes focused on the cost-volume kernel (one frame each): issue mix and texture-address unit
cd /root/repo; export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/cv_pmc; mkdir -p $O
cd /tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
D="TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"
i=0
for C_ in "$A" "$D"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $C_ -d /tmp/pmc_$i -- python $R/tools/prof_render.py 1 > $O/pmc_$i.log 2>&1
done
python $R/tools/pmc_summary.py /tmp > $O/pmc_summary.txt 2>&1
grep -A18 "cost_volume" $O/pmc_summary.txt | head -20
