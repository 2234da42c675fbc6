#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export PYTHONPATH=/root/repo
timeout 1200 python -m pytest tests/test_stress_gpu.py tests/test_model_gpu.py tests/test_hip_kernels.py tests/test_encoder_layer_backward.py tests/test_decoder_backward.py tests/test_window_attention_backward.py tests/test_dist_gpu.py -q > gpurun_out/r4r_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4r_tests.log
timeout 300 python tools/exp/wa_bwd_time.py > gpurun_out/r4r_train.log 2>&1
tail -5 gpurun_out/r4r_tests.log; grep -v Warn gpurun_out/r4r_train.log | grep "ms per" | head -4
