#!/bin/bash
# PMC pass over the window-attention backward alone (matrix-pipe busy fraction of the split-bf16 kernels)
R=/root/repo; cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
rm -rf /tmp/pmc_wab; mkdir -p /tmp/pmc_wab
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY -d /tmp/pmc_wab/a -- python $R/tools/exp/wa_bwd_alone.py > $R/gpurun_out/r4x_pmc.log 2>&1
python $R/tools/pmc_summary.py /tmp/pmc_wab > $R/gpurun_out/r4x_pmc_summary.txt 2>&1
grep -A9 "wa_bwd" $R/gpurun_out/r4x_pmc_summary.txt | head -60
