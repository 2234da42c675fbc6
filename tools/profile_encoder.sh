#!/bin/bash
# PMC counters of the encoder's kernels (one steady-state pass pair of tools/prof_encoder.py), ON THE GPU BOX:
#   tools/profile_encoder.sh r2_07  ->  gpurun_out/<tag>/encoder_pmc_summary.txt
# Counters in separate rocprofv3 runs, kernel trace only (no sys / hip tracing with --pmc).
TAG=${1:-enc}
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; rm -rf /tmp/pmce_${TAG}*
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
B="SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
for P in a b; do
  case $P in a) C="$A";; b) C="$B";; esac
  timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc $C -d /tmp/pmce_${TAG}_$P -- python $R/tools/prof_encoder.py 1 > $O/enc_pmc_$P.log 2>&1
done
mkdir -p /tmp/pmce_${TAG}; mv /tmp/pmce_${TAG}_? /tmp/pmce_${TAG}/ 2>/dev/null
python $R/tools/pmc_summary.py /tmp/pmce_${TAG} > $O/encoder_pmc_summary.txt 2>&1
grep -c "^##" $O/encoder_pmc_summary.txt
