#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export PYTHONPATH=/root/repo
rm -f gpurun_out/r4w.log
for m in "" 0 100000; do
if [ -z "$m" ]; then timeout 100 python tools/exp/frame_time.py c2 8 2>&1 | grep frame >> gpurun_out/r4w.log
else MNERF_WA_MIN4=$m timeout 100 python tools/exp/frame_time.py c2 8 2>&1 | grep frame >> gpurun_out/r4w.log; fi
done
timeout 120 python tools/exp/video_time.py 64 80 30 5 2>&1 | grep poses | tail -2 >> gpurun_out/r4w.log
cat gpurun_out/r4w.log
