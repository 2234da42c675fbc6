#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export PYTHONPATH=/root/repo
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_encoder_layer_backward.py tests/test_decoder_backward.py tests/test_window_attention_backward.py -x -q > gpurun_out/r4m_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4m_tests.log
timeout 300 python tools/exp/gemm_time.py > gpurun_out/r4m_gemm.log 2>&1
timeout 300 python tools/exp/wa_bwd_time.py > gpurun_out/r4m_train.log 2>&1
MNERF_GEMM_MATH=f32 timeout 300 python tools/exp/wa_bwd_time.py >> gpurun_out/r4m_train.log 2>&1
tail -4 gpurun_out/r4m_tests.log; cat gpurun_out/r4m_gemm.log | grep -v Warn; grep -v Warn gpurun_out/r4m_train.log | tail -12
