#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export PYTHONPATH=/root/repo
rm -f gpurun_out/r4v_train.log
for i in 1 2; do
timeout 300 python tools/exp/wa_bwd_time.py 2>&1 | grep "ms per" | head -1 >> gpurun_out/r4v_train.log
MNERF_WA_FWD_STATS=0 timeout 300 python tools/exp/wa_bwd_time.py 2>&1 | grep "ms per" | head -1 | sed 's/^/no forward statistics: /' >> gpurun_out/r4v_train.log
done
cat gpurun_out/r4v_train.log
