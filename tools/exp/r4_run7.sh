#!/bin/bash
# round-4 GPU call 7: transformer-layer backward in HIP (tests, train-step timing and trace), the suite
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/${1:-r4g}; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_encoder_layer_backward.py tests/test_window_attention_backward.py -m gpu -q -x -s > $O/pytest_enc.log 2>&1; echo "encoder backward tests rc=$?" | tee -a $O/summary.txt; grep -E "passed|failed|Error|\{" $O/pytest_enc.log | tail -14 | cut -c1-400 | tee -a $O/summary.txt
timeout 600 python tools/exp/wa_bwd_time.py > $O/wa_bwd_time.log 2>&1; echo "wa_bwd_time rc=$?" | tee -a $O/summary.txt; grep -E "ms per" $O/wa_bwd_time.log | tee -a $O/summary.txt
cd /tmp; rm -rf /tmp/prof_train; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_train -o trace -- python $R/tools/exp/train_step_prof.py 3 > $O/train_trace.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/prof_train -name '*.db' | head -1) 40 > $O/train_kernel_stats.md 2>&1; head -34 $O/train_kernel_stats.md | cut -c1-170 | tee -a $O/summary.txt
cd $R
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_dist_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt; tail -4 $O/pytest.log | tee -a $O/summary.txt
