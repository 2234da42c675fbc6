#!/bin/bash
# One parameterised GPU call instead of a script per call (the round-4 r4_run*.sh notebook is in the history):
#   gpurun -- 'bash tools/exp/gpu_call.sh TAG step [step ...]'      -> everything under gpurun_out/TAG/
# steps:
#   tests[=EXPR]        python -m pytest tests -m gpu -q [-k EXPR]            -> tests.log
#   bench[=STEPS]       python bench.py --steps STEPS (default 10)            -> bench.json, key figures on stdout
#   frames=C1,C2,..     tools/exp/frame_time.py per configuration (c2 c3 c5 s<N> demo); MNERF_LIB_VARIANTS="a b" runs every
#                       configuration with libmnerf_hip.so and libmnerf_hip_<a>.so ... in turn (same-box A/B)  -> frames.log
#   profile             tools/profile_round.sh TAG (kernel stats, bench trace, PMC passes, counter json files)
#   config=CFG          tools/profile_config.sh TAG_CFG CFG (kernel stats + PMC of one secondary configuration: c3 c5 s256 ...)
#   trace=SCRIPT[,ARG..] rocprofv3 --kernel-trace --stats over python tools/exp/SCRIPT ARGS -> SCRIPT_kernel_stats.md
#                       (TRACE_PERIODS=K: only the last K periods between decoder launches, e.g. training iterations)
#   py=SCRIPT[,ARG..]   python tools/exp/SCRIPT ARGS                           -> SCRIPT.log
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
for step in "$@"; do
  name=${step%%=*}; arg=""; [[ $step == *=* ]] && arg=${step#*=}
  case $name in
    tests)
      timeout 1800 python -m pytest tests -m gpu -q ${arg:+-k "$arg"} > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
      tail -4 $O/tests.log;;
    bench)
      timeout 900 python bench.py --steps ${arg:-10} --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
      python - "$O/bench.json" <<'PY'
import json, sys
l = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = l["roofline"]
print("value", l["value"], "ms_per_step", l["ms_per_step"], "frac", r["frac"], "traffic", r.get("traffic"))
print("cost_volume", json.dumps(r.get("cost_volume"))[:900])
print("frame", json.dumps({k: v for k, v in r.get("frame", {}).items() if k != "what"}))
for w in l["config"].get("secondary_workloads") or []:
    print({k: w.get(k) for k in ("workload", "ms_per_frame", "encoder_ms", "cost_volume_ms", "decoder_ms", "ms_per_iteration", "sample_intvs", "error") if k in w})
print("cpu_baseline", l.get("cpu_baseline", {}).get("value"))
PY
      ;;
    frames)
      for rep in 1 2; do for v in "" $MNERF_LIB_VARIANTS; do for c in ${arg//,/ }; do
        MNERF_LIB=$PWD/matchnerf_amd/libmnerf_hip${v:+_$v}.so timeout 300 python tools/exp/frame_time.py $c 6 2>&1 | tail -1 | sed "s/^/[${v:-base}] /" | tee -a $O/frames.log
      done; done; done;;
    profile)
      bash tools/profile_round.sh $TAG; ls $O;;
    config)
      bash tools/profile_config.sh ${TAG}_$arg $arg | cut -c1-170;;
    trace)
      IFS=, read -r script rest <<< "$arg"
      R=$PWD; rm -rf /tmp/trace_$TAG; ( cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/trace_$TAG -o t -- python $R/tools/exp/$script ${rest//,/ } > $R/$O/${script%.py}_trace.log 2>&1 )
      python tools/rocpd_stats.py $(find /tmp/trace_$TAG -name '*.db' | head -1) 45 ${TRACE_PERIODS:+--periods $TRACE_PERIODS} > $O/${script%.py}_kernel_stats.md 2>&1
      head -14 $O/${script%.py}_kernel_stats.md | cut -c1-160;;
    py)
      IFS=, read -r script rest <<< "$arg"
      timeout 900 python tools/exp/$script ${rest//,/ } 2>&1 | grep -v "amdgpu.ids" | tee $O/${script%.py}.log;;
    *) echo "unknown step $step";;
  esac
done
