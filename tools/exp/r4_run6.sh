#!/bin/bash
# round-4 GPU call 6: DPP compositing (fixed), attention backward with prefetch, pp decoder at 7 / 10 views, train-step profile
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/${1:-r4f}; mkdir -p $O
export TMPDIR=/tmp
ft() { # name lib [cfg]
  if [ -n "$2" ]; then MNERF_LIB=$R/matchnerf_amd/libmnerf_hip_$2.so timeout 300 python tools/exp/frame_time.py ${3:-c2} 6 > $O/ft_$1.log 2>&1; else timeout 300 python tools/exp/frame_time.py ${3:-c2} 6 > $O/ft_$1.log 2>&1; fi
  echo "$1: $(tail -1 $O/ft_$1.log)" | tee -a $O/summary.txt
}
ft main1 ""; ft t4dpp t4dpp; ft r3dec r3dec; ft main2 ""; ft t4dpp2 t4dpp
MNERF_LIB=$R/matchnerf_amd/libmnerf_hip_t4dpp.so timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_model_gpu.py tests/test_stress_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x > $O/pytest_t4dpp.log 2>&1; echo "t4dpp tests rc=$?" | tee -a $O/summary.txt; tail -3 $O/pytest_t4dpp.log | tee -a $O/summary.txt
ft c5 "" c5; MNERF_DECODER_PP=0 timeout 300 python tools/exp/frame_time.py c5 3 > $O/ft_c5_nopp.log 2>&1; echo "c5 nopp: $(tail -1 $O/ft_c5_nopp.log)" | tee -a $O/summary.txt
timeout 600 python -m pytest tests/test_window_attention_backward.py -m gpu -q -x > $O/pytest_wab.log 2>&1; echo "wa backward tests rc=$?" | tee -a $O/summary.txt; tail -2 $O/pytest_wab.log | tee -a $O/summary.txt
timeout 600 python tools/exp/wa_bwd_time.py > $O/wa_bwd_time.log 2>&1; echo "wa_bwd_time rc=$?" | tee -a $O/summary.txt; grep -E "ms per" $O/wa_bwd_time.log | tee -a $O/summary.txt
cd /tmp; rm -rf /tmp/prof_train; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_train -o trace -- python $R/tools/exp/train_step_prof.py 3 > $O/train_trace.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/prof_train -name '*.db' | head -1) 45 > $O/train_kernel_stats.md 2>&1; head -60 $O/train_kernel_stats.md | tee -a $O/summary.txt
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt; tail -4 $O/pytest.log | tee -a $O/summary.txt
