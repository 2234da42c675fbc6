#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export PYTHONPATH=/root/repo
timeout 900 python -m pytest tests/test_window_attention_backward.py tests/test_encoder_layer_backward.py tests/test_model_gpu.py tests/test_hip_kernels.py -q > gpurun_out/r4u_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4u_tests.log
timeout 300 python tools/exp/wa_bwd_time.py 2>&1 | grep "ms per" | head -1 > gpurun_out/r4u_train.log
timeout 200 python tools/exp/frame_time.py c2 6 2>&1 | grep frame >> gpurun_out/r4u_train.log
tail -4 gpurun_out/r4u_tests.log; cat gpurun_out/r4u_train.log
