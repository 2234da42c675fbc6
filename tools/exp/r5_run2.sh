#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r5_2; mkdir -p $O
python tools/exp/enc_err.py 2>&1 | grep -v Warn | tee $O/enc_err.log
MNERF_WA_MATH=f32 python tools/exp/enc_err.py 2>&1 | grep -v Warn | tee -a $O/enc_err.log
timeout 900 python -m pytest tests -m gpu -q -k "demo_own or real_scene or scene" > $O/tests_demo.log 2>&1; echo "tests rc=$?" >> $O/tests_demo.log
tail -8 $O/tests_demo.log
