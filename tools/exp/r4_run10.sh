#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export PYTHONPATH=/root/repo
rm -f gpurun_out/r4j_frame.log
for n in 65536 81920 163840 327680; do
MNERF_CV_GRID=1000000 MNERF_MAX_RAYS_PER_LAUNCH=$n timeout 200 python tools/exp/size_time.py 512 640 5 >> gpurun_out/r4j_frame.log 2>&1
done
MNERF_CV_GRID=1000000 timeout 200 python tools/exp/frame_time.py c5 3 >> gpurun_out/r4j_frame.log 2>&1
timeout 200 python tools/exp/frame_time.py c5 3 >> gpurun_out/r4j_frame.log 2>&1
MNERF_CV_GRID=1000000 timeout 200 python tools/exp/frame_time.py c3 3 >> gpurun_out/r4j_frame.log 2>&1
timeout 200 python tools/exp/frame_time.py c3 3 >> gpurun_out/r4j_frame.log 2>&1
grep frame gpurun_out/r4j_frame.log
