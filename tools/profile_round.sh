#!/bin/bash
# One command that regenerates every profile-derived number of a round ON THE GPU BOX:
#   tools/profile_round.sh r2_01
# -> gpurun_out/<tag>/{kernel_stats.md, bench_kernel_stats.md, pmc_summary.txt, decoder_counters.json, cost_volume_counters.json}; copy
# the ones to be judged into profiles/ (profiles/current/<tag>_kernel_stats.md, profiles/current/<tag>_pmc_summary.txt, profiles/current/decoder_counters.json,
# profiles/current/cost_volume_counters.json).
# Kernel trace and PMC counters are collected in SEPARATE rocprofv3 runs (no sys / hip tracing with --pmc).
TAG=${1:-r2}
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; rm -rf /tmp/prof_$TAG /tmp/pmc_${TAG}_*
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o trace -- python $R/tools/prof_render.py 3 > $O/trace.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/prof_$TAG -name '*.db' | head -1) 40 --last-frame 5 > $O/kernel_stats.md 2>&1
# the bench command itself under the kernel trace (whole run: warm-up, timed steps, the other-math and secondary legs)
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_${TAG}_bench -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_traced.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/prof_${TAG}_bench -name '*.db' | head -1) 25 > $O/bench_kernel_stats.md 2>&1
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
B="SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT"
E="TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"  # cost volume: which unit binds it
for P in a b c d e; do
  case $P in a) C="$A";; b) C="$B";; c) C="FETCH_SIZE TCC_HIT_sum TCP_TCC_READ_REQ_sum";; d) C="WRITE_SIZE TCC_MISS_sum TCC_EA0_RDREQ_sum";; e) C="$E";; esac
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $C -d /tmp/pmc_${TAG}_$P -- python $R/tools/prof_render.py 1 > $O/pmc_$P.log 2>&1
done
mkdir -p /tmp/pmc_${TAG}; mv /tmp/pmc_${TAG}_? /tmp/pmc_${TAG}/ 2>/dev/null
python $R/tools/pmc_summary.py /tmp/pmc_${TAG} > $O/pmc_summary.txt 2>&1
python $R/tools/pmc_to_json.py /tmp/pmc_${TAG} 1 $O/decoder_counters.json > $O/pmc_to_json.log 2>&1
tail -2 $O/pmc_to_json.log
