#!/bin/bash
# round 5, call 1: the real-scene parity tests, S sweep of the frame, cost-volume PMC passes (issue mix + texture-address unit)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r5_1; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "demo_own or real_scene or scene" > $O/tests_demo.log 2>&1; echo "tests rc=$?" >> $O/tests_demo.log
tail -15 $O/tests_demo.log
for c in c2 s128 s256 demo; do timeout 300 python tools/exp/frame_time.py $c 4 2>&1 | tail -1 | tee -a $O/frames.log; done
timeout 600 bash tools/exp/prof_cv.sh > $O/prof_cv.log 2>&1
cp gpurun_out/cv_pmc/pmc_summary.txt $O/cv_pmc_summary.txt
grep -B2 -A30 "cost_volume" $O/cv_pmc_summary.txt | head -60
