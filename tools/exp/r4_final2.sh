#!/bin/bash
# round-4 closing call: GPU suite, smoke, bench (decoder counters replayed from profiles/decoder_counters.json), the bench command and three
# training steps under the kernel trace
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; T=${1:-r4_03}; O=$R/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=$R
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt; tail -4 $O/pytest.log | tee -a $O/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt; tail -2 $O/smoke.log | tee -a $O/summary.txt
( time timeout 400 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; echo "bench rc=$?" | tee -a $O/summary.txt; tail -3 $O/bench.time | tee -a $O/summary.txt
python -c "
import json;d=json.load(open('$O/bench.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['traffic'],d['roofline'].get('mfma_busy_measured'),d['config']['decoder_ms_per_frame'],d['config']['cost_volume_ms_per_frame'],d['config']['encoder_ms']); print([ (w.get('workload','')[:24], w.get('ms_per_frame', w.get('ms_per_iteration')), w.get('decoder_ms'), w.get('cost_volume_ms'), w.get('encoder_ms')) for w in d['config']['secondary_workloads']])" | tee -a $O/summary.txt
cd /tmp; rm -rf /tmp/prof_$T; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$T -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_traced.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/prof_$T -name '*.db' | head -1) 25 > $O/bench_kernel_stats.md 2>&1
rm -rf /tmp/train_$T; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/train_$T -o t -- python $R/tools/exp/train_step_prof.py 3 > $O/train_prof.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/train_$T -name '*.db' | head -1) 40 > $O/train_kernel_stats.md 2>&1
head -16 $O/train_kernel_stats.md | cut -c1-150 | tee -a $O/summary.txt
