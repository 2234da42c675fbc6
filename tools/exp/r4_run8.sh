#!/bin/bash
# round-4 GPU call 8: 128x128-tile f32 GEMM (decoder / encoder backward tests, train-step timing and trace)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/${1:-r4h}; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_encoder_layer_backward.py tests/test_decoder_backward.py tests/test_window_attention_backward.py -m gpu -q -x > $O/pytest_bwd.log 2>&1; echo "backward tests rc=$?" | tee -a $O/summary.txt; tail -3 $O/pytest_bwd.log | tee -a $O/summary.txt
timeout 600 python tools/exp/wa_bwd_time.py > $O/wa_bwd_time.log 2>&1; echo "wa_bwd_time rc=$?" | tee -a $O/summary.txt; grep -E "ms per" $O/wa_bwd_time.log | tee -a $O/summary.txt
timeout 300 python tools/exp/backward_time.py > $O/backward_time.log 2>&1; tail -4 $O/backward_time.log | tee -a $O/summary.txt
cd /tmp; rm -rf /tmp/prof_train; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_train -o trace -- python $R/tools/exp/train_step_prof.py 3 > $O/train_trace.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/prof_train -name '*.db' | head -1) 40 > $O/train_kernel_stats.md 2>&1; head -24 $O/train_kernel_stats.md | cut -c1-170 | tee -a $O/summary.txt
cd $R
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt; tail -3 $O/pytest.log | tee -a $O/summary.txt
