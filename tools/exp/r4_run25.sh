#!/bin/bash
# PMC passes over tools/exp/gemm_time.py (the split-bf16 GEMM: matrix-pipe busy fraction, bytes fetched past L2)
R=/root/repo; cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
rm -rf /tmp/pmc_gemm; mkdir -p /tmp/pmc_gemm
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS GRBM_GUI_ACTIVE -d /tmp/pmc_gemm/a -- python $R/tools/exp/gemm_time.py > $R/gpurun_out/r4y_pmc_a.log 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum -d /tmp/pmc_gemm/c -- python $R/tools/exp/gemm_time.py > $R/gpurun_out/r4y_pmc_c.log 2>&1
python $R/tools/pmc_summary.py /tmp/pmc_gemm > $R/gpurun_out/r4y_pmc_summary.txt 2>&1
grep -A12 "gemm_b6" $R/gpurun_out/r4y_pmc_summary.txt | head -80
