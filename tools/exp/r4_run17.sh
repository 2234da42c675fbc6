#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export PYTHONPATH=/root/repo
timeout 600 python -m pytest tests/test_hip_kernels.py -x -q -k "cost_volume_backward" > gpurun_out/r4q_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4q_tests.log
rm -f gpurun_out/r4q_time.log
timeout 200 python tools/exp/cvb_time.py >> gpurun_out/r4q_time.log 2>&1
MNERF_CV_BWD_WALK=0 timeout 200 python tools/exp/cvb_time.py >> gpurun_out/r4q_time.log 2>&1
MNERF_LIB=/root/repo/matchnerf_amd/libmnerf_hip_cvb1.so timeout 200 python tools/exp/cvb_time.py >> gpurun_out/r4q_time.log 2>&1
tail -3 gpurun_out/r4q_tests.log; grep "ms per call" gpurun_out/r4q_time.log
