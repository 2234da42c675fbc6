#!/bin/bash
# round-4 measurement call: GPU suite, bench, kernel stats + PMC of config[1] (profile_round), of config[4] and config[2], phase timeline
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; T=${1:-r4_01}; O=$R/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt; tail -4 $O/pytest.log | tee -a $O/summary.txt
( time timeout 400 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; echo "bench rc=$?" | tee -a $O/summary.txt; tail -3 $O/bench.time | tee -a $O/summary.txt
python -c "
import json;d=json.load(open('$O/bench.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['config']['decoder_ms_per_frame'],d['config']['cost_volume_ms_per_frame'],d['config']['encoder_ms']); print([ (w.get('workload','')[:24], w.get('ms_per_frame', w.get('ms_per_iteration')), w.get('decoder_ms'), w.get('cost_volume_ms'), w.get('encoder_ms')) for w in d['config']['secondary_workloads']])" | tee -a $O/summary.txt
MNERF_LIB=$R/matchnerf_amd/libmnerf_hip_tl.so timeout 200 python tools/exp/pp_timeline.py > $O/pp_timeline.log 2>&1; echo "timeline rc=$?" | tee -a $O/summary.txt
bash tools/profile_round.sh $T > $O/profile_round.log 2>&1; echo "profile_round rc=$?" | tee -a $O/summary.txt; head -30 $O/kernel_stats.md | tee -a $O/summary.txt; cat $O/pmc_to_json.log | tail -3 | tee -a $O/summary.txt
bash tools/profile_config.sh ${T%_*}_c5 c5 > $R/gpurun_out/${T%_*}_c5.log 2>&1; echo "profile c5 rc=$?" | tee -a $O/summary.txt
bash tools/profile_config.sh ${T%_*}_c3 c3 > $R/gpurun_out/${T%_*}_c3.log 2>&1; echo "profile c3 rc=$?" | tee -a $O/summary.txt
