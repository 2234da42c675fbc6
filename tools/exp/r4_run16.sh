#!/bin/bash
cd /tmp; export TMPDIR=/tmp; export PYTHONPATH=/root/repo
rm -rf /tmp/train_prof; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/train_prof -o t -- python /root/repo/tools/exp/train_step_prof.py 3 > /root/repo/gpurun_out/r4p_prof.log 2>&1
python /root/repo/tools/rocpd_stats.py $(find /tmp/train_prof -name '*.db' | head -1) 45 > /root/repo/gpurun_out/r4p_train_kernel_stats.md 2>&1
head -40 /root/repo/gpurun_out/r4p_train_kernel_stats.md | cut -c1-140
