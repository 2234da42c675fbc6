#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export PYTHONPATH=/root/repo
rm -f gpurun_out/r4l_frame.log
for i in 1 2 3; do
for v in "" km0 km2; do
if [ -z "$v" ]; then timeout 200 python tools/exp/frame_time.py c2 6 >> gpurun_out/r4l_frame.log 2>&1
else MNERF_LIB=/root/repo/matchnerf_amd/libmnerf_hip_$v.so timeout 200 python tools/exp/frame_time.py c2 6 >> gpurun_out/r4l_frame.log 2>&1; fi
done
done
timeout 200 python tools/exp/frame_time.py c3 3 >> gpurun_out/r4l_frame.log 2>&1
MNERF_LIB=/root/repo/matchnerf_amd/libmnerf_hip_km0.so timeout 200 python tools/exp/frame_time.py c3 3 >> gpurun_out/r4l_frame.log 2>&1
grep frame gpurun_out/r4l_frame.log
