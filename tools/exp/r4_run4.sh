#!/bin/bash
# round-4 GPU call 4: T2 without the padded-key selects, M-loop fetch variants, DPP compositing, encoder graph; suite; bench
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/${1:-r4d}; mkdir -p $O
export TMPDIR=/tmp
ft() { # name lib
  if [ -n "$2" ]; then MNERF_LIB=$R/matchnerf_amd/libmnerf_hip_$2.so timeout 200 python tools/exp/frame_time.py c2 6 > $O/ft_$1.log 2>&1; else timeout 200 python tools/exp/frame_time.py c2 6 > $O/ft_$1.log 2>&1; fi
  echo "$1: $(tail -1 $O/ft_$1.log)" | tee -a $O/summary.txt
}
ft main1 ""; ft r3dec r3dec; ft depth3 depth3; ft group2 group2; ft t4dpp t4dpp; ft main2 ""
MNERF_ENCODER_GRAPH=0 timeout 200 python tools/exp/frame_time.py c2 6 > $O/ft_nograph.log 2>&1; echo "nograph: $(tail -1 $O/ft_nograph.log)" | tee -a $O/summary.txt
MNERF_LIB=$R/matchnerf_amd/libmnerf_hip_tl.so timeout 200 python tools/exp/pp_timeline.py > $O/pp_timeline.log 2>&1; echo "timeline rc=$?" | tee -a $O/summary.txt
timeout 200 python tools/exp/frame_time.py c3 2 > $O/ft_c3.log 2>&1; tail -1 $O/ft_c3.log | tee -a $O/summary.txt
timeout 300 python tools/exp/frame_time.py c5 2 > $O/ft_c5.log 2>&1; tail -1 $O/ft_c5.log | tee -a $O/summary.txt
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt; tail -5 $O/pytest.log | tee -a $O/summary.txt
( time timeout 400 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; echo "bench rc=$?" | tee -a $O/summary.txt
python -c "
import json;d=json.load(open('$O/bench.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['config']['decoder_ms_per_frame'],d['config']['cost_volume_ms_per_frame'],d['config']['encoder_ms'])" | tee -a $O/summary.txt
