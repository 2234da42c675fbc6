#!/bin/bash
# kernel trace + one PMC pass (clock / MFMA busy) for the render kernels of the current build
cd /root/repo; export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/dec_pmc; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof -o r1d -- python $R/tools/prof_render.py 3 > $O/trace.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/prof -name '*.db' | head -1) 2>&1 | grep -E "decoder_kernel|cost_volume|window_attention|kernel stats" > $O/kernel_stats.md
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $A -d /tmp/pmc_a -- python $R/tools/prof_render.py 1 > $O/pmc_a.log 2>&1
python $R/tools/pmc_summary.py /tmp > $O/pmc_summary.txt 2>&1
cat $O/kernel_stats.md; grep -A12 "decoder_kernel" $O/pmc_summary.txt | head -14
