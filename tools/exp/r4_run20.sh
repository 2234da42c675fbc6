#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export PYTHONPATH=/root/repo
timeout 600 python -m pytest tests/test_encoder_layer_backward.py tests/test_model_gpu.py -q -k "backward or gradients or layer" > gpurun_out/r4t_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4t_tests.log
timeout 300 python tools/exp/wa_bwd_time.py 2>&1 | grep "ms per" | head -2 > gpurun_out/r4t_train.log
tail -3 gpurun_out/r4t_tests.log; cat gpurun_out/r4t_train.log
