#!/bin/bash
# Kernel stats + PMC counters of ONE secondary configuration ON THE GPU BOX:  tools/profile_config.sh r3_c3 c3 | r3_c5 c5
# -> gpurun_out/<tag>/{kernel_stats.md, pmc_summary.txt}  (copy into profiles/<tag>_*).  Separate rocprofv3 runs for the
# kernel trace and for every counter group (no sys / hip tracing together with --pmc).
TAG=${1:-r3_c3}; CFG=${2:-c3}
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; rm -rf /tmp/prof_$TAG /tmp/pmc_${TAG}*
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o trace -- python $R/tools/prof_render.py 2 $CFG > $O/trace.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/prof_$TAG -name '*.db' | head -1) 30 > $O/kernel_stats.md 2>&1
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
B="SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT TA_TA_BUSY_sum TA_BUSY_avr"
mkdir -p /tmp/pmc_${TAG}
for P in a b c d; do
  case $P in a) C="$A";; b) C="$B";; c) C="FETCH_SIZE TCC_HIT_sum TCP_TCC_READ_REQ_sum";; d) C="WRITE_SIZE TCC_MISS_sum TCC_EA0_RDREQ_sum";; esac
  timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc $C -d /tmp/pmc_${TAG}/$P -- python $R/tools/prof_render.py 1 $CFG > $O/pmc_$P.log 2>&1
done
python $R/tools/pmc_summary.py /tmp/pmc_${TAG} > $O/pmc_summary.txt 2>&1
head -40 $O/kernel_stats.md
