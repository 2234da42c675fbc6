#!/bin/bash
# round-4 GPU call 2: decoder variants (spread / burst requests, team-B priority, split form), timeline, pair-blocked cost volume, suite, bench
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/${1:-r4b}; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
timeout 200 python tools/exp/frame_time.py c2 4 > $O/ft_main.log 2>&1; tail -1 $O/ft_main.log | tee -a $O/summary.txt
for v in burst nobprio cvt; do
  MNERF_LIB=$R/matchnerf_amd/libmnerf_hip_$v.so timeout 200 python tools/exp/frame_time.py c2 4 > $O/ft_$v.log 2>&1; tail -1 $O/ft_$v.log | tee -a $O/summary.txt
done
MNERF_DECODER_PP=0 timeout 200 python tools/exp/frame_time.py c2 4 > $O/ft_nopp.log 2>&1; tail -1 $O/ft_nopp.log | tee -a $O/summary.txt
MNERF_LIB=$R/matchnerf_amd/libmnerf_hip_tl.so timeout 200 python tools/exp/pp_timeline.py > $O/pp_timeline.log 2>&1; echo "timeline rc=$?" | tee -a $O/summary.txt
timeout 200 python tools/exp/frame_time.py c3 2 > $O/ft_c3.log 2>&1; tail -1 $O/ft_c3.log | tee -a $O/summary.txt
timeout 300 python tools/exp/frame_time.py c5 2 > $O/ft_c5.log 2>&1; tail -1 $O/ft_c5.log | tee -a $O/summary.txt
MNERF_CV_PAIR_BLOCK=0 timeout 300 python tools/exp/frame_time.py c5 2 > $O/ft_c5_noblk.log 2>&1; tail -1 $O/ft_c5_noblk.log | tee -a $O/summary.txt
MNERF_CV_PAIR_BLOCK=4 timeout 300 python tools/exp/frame_time.py c5 2 > $O/ft_c5_blk4.log 2>&1; tail -1 $O/ft_c5_blk4.log | tee -a $O/summary.txt
MNERF_CV_PAIR_BLOCK=15 timeout 300 python tools/exp/frame_time.py c5 2 > $O/ft_c5_blk15.log 2>&1; tail -1 $O/ft_c5_blk15.log | tee -a $O/summary.txt
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt; tail -5 $O/pytest.log | tee -a $O/summary.txt
( time timeout 400 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; echo "bench rc=$?" | tee -a $O/summary.txt; cat $O/bench.time | tee -a $O/summary.txt
python -c "
import json;d=json.load(open('$O/bench.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['config']['decoder_ms_per_frame'],d['config']['cost_volume_ms_per_frame'],d['config']['encoder_ms'])" | tee -a $O/summary.txt
