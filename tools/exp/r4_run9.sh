#!/bin/bash
# pose table: tests, video timing, sanity frame time
cd /root/repo
mkdir -p gpurun_out
export PYTHONPATH=/root/repo
timeout 900 python -m pytest tests/test_pose_table_gpu.py tests/test_model_gpu.py::test_video_mode_renders_each_pose -x -q > gpurun_out/r4i_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4i_tests.log
echo > gpurun_out/r4i_video.log
rm -f gpurun_out/r4i_frame.log
for i in 1 2 3; do
timeout 200 python tools/exp/frame_time.py c2 6 >> gpurun_out/r4i_frame.log 2>&1
MNERF_EXP_SINGLE_TABLE=1 timeout 200 python tools/exp/frame_time.py c2 6 >> gpurun_out/r4i_frame.log 2>&1
done
tail -5 gpurun_out/r4i_tests.log; cat gpurun_out/r4i_video.log | grep -v Warn; cat gpurun_out/r4i_frame.log | grep frame
