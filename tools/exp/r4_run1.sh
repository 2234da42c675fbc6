#!/bin/bash
# round-4 GPU call 1: smoke, the GPU suite, decoder variants, phase timeline, bench.  Every step under its own timeout.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/${1:-r4a}; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
timeout 200 python tools/exp/frame_time.py c2 4 > $O/ft_main.log 2>&1; echo "frame_time main rc=$?" | tee -a $O/summary.txt; tail -1 $O/ft_main.log | tee -a $O/summary.txt
for v in burst prio1 prio3; do
  MNERF_LIB=$R/matchnerf_amd/libmnerf_hip_$v.so timeout 200 python tools/exp/frame_time.py c2 4 > $O/ft_$v.log 2>&1; echo "frame_time $v rc=$?" | tee -a $O/summary.txt; tail -1 $O/ft_$v.log | tee -a $O/summary.txt
done
MNERF_DECODER_PP=0 timeout 200 python tools/exp/frame_time.py c2 4 > $O/ft_nopp.log 2>&1; tail -1 $O/ft_nopp.log | tee -a $O/summary.txt
MNERF_LIB=$R/matchnerf_amd/libmnerf_hip_tl.so timeout 200 python tools/exp/pp_timeline.py > $O/pp_timeline.log 2>&1; echo "timeline rc=$?" | tee -a $O/summary.txt
timeout 200 python tools/exp/frame_time.py c3 2 > $O/ft_c3.log 2>&1; tail -1 $O/ft_c3.log | tee -a $O/summary.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt; tail -3 $O/pytest.log | tee -a $O/summary.txt
( time timeout 400 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; echo "bench rc=$?" | tee -a $O/summary.txt; cat $O/bench.time | tee -a $O/summary.txt
python -c "
import json;d=json.load(open('$O/bench.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['config']['decoder_ms_per_frame'],d['config']['cost_volume_ms_per_frame'],d['config']['encoder_ms'])" | tee -a $O/summary.txt
