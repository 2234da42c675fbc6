#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export PYTHONPATH=/root/repo
timeout 600 python -m pytest tests/test_window_attention_backward.py -x -q -s > gpurun_out/r4n_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4n_tests.log
rm -f gpurun_out/r4n_alone.log
timeout 100 python tools/exp/wa_bwd_alone.py >> gpurun_out/r4n_alone.log 2>&1
MNERF_WA_BWD_MATH=f32 timeout 100 python tools/exp/wa_bwd_alone.py >> gpurun_out/r4n_alone.log 2>&1
for v in $WB_VARIANTS; do MNERF_LIB=/root/repo/matchnerf_amd/libmnerf_hip_$v.so timeout 100 python tools/exp/wa_bwd_alone.py >> gpurun_out/r4n_alone.log 2>&1; done
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/wab_prof; timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/wab_prof -o t -- python /root/repo/tools/exp/wa_bwd_alone.py > /dev/null 2>&1
python /root/repo/tools/rocpd_stats.py $(find /tmp/wab_prof -name '*.db' | head -1) 8 > /root/repo/gpurun_out/r4n_kernels.md 2>&1
cd /root/repo; grep -v "^$" gpurun_out/r4n_tests.log | tail -10; grep "ms per call" gpurun_out/r4n_alone.log; grep "wa_bwd\|attention" gpurun_out/r4n_kernels.md | cut -c1-150
