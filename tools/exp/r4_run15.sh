#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export PYTHONPATH=/root/repo
timeout 900 python -m pytest tests/test_window_attention_backward.py tests/test_encoder_layer_backward.py tests/test_model_gpu.py -x -q > gpurun_out/r4o_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4o_tests.log
timeout 300 python tools/exp/wa_bwd_time.py > gpurun_out/r4o_train.log 2>&1
tail -4 gpurun_out/r4o_tests.log; grep -v Warn gpurun_out/r4o_train.log | grep "ms per"
