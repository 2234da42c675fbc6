#!/bin/bash
# round-4 GPU call 5: window-attention backward (tests, timing), DPP compositing (row ops only), padded cost-volume LDS
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/${1:-r4e}; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_window_attention_backward.py -m gpu -q -x -s > $O/pytest_wab.log 2>&1; echo "wa backward tests rc=$?" | tee -a $O/summary.txt; grep -E "passed|failed|\{" $O/pytest_wab.log | tail -12 | tee -a $O/summary.txt
ft() { # name lib
  if [ -n "$2" ]; then MNERF_LIB=$R/matchnerf_amd/libmnerf_hip_$2.so timeout 200 python tools/exp/frame_time.py c2 6 > $O/ft_$1.log 2>&1; else timeout 200 python tools/exp/frame_time.py c2 6 > $O/ft_$1.log 2>&1; fi
  echo "$1: $(tail -1 $O/ft_$1.log)" | tee -a $O/summary.txt
}
ft main1 ""; ft t4dpp t4dpp; ft cvnopad cvnopad; ft main2 ""; ft t4dpp2 t4dpp; ft cvnopad2 cvnopad
MNERF_LIB=$R/matchnerf_amd/libmnerf_hip_t4dpp.so timeout 600 python -m pytest tests/test_hip_kernels.py tests/test_model_gpu.py tests/test_stress_gpu.py -m gpu -q -x > $O/pytest_t4dpp.log 2>&1; echo "t4dpp tests rc=$?" | tee -a $O/summary.txt; tail -3 $O/pytest_t4dpp.log | tee -a $O/summary.txt
timeout 600 python tools/exp/wa_bwd_time.py > $O/wa_bwd_time.log 2>&1; echo "wa_bwd_time rc=$?" | tee -a $O/summary.txt; grep -E "ms per" $O/wa_bwd_time.log | tee -a $O/summary.txt
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_hip_kernels.py tests/test_fullsize_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt; tail -4 $O/pytest.log | tee -a $O/summary.txt
