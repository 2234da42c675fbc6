#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export PYTHONPATH=/root/repo
rm -f gpurun_out/r4s_gemm.log
for cfg in "0 128 1024" "128 128 1024" "128 128 256" "128 256 256" "64 256 1024" "64 512 1024" "64 128 2048" "64 256 2048"; do set -- $cfg
echo "== tile $1 min_k $2 wgs $3" >> gpurun_out/r4s_gemm.log
MNERF_GEMM_FORCE_TILE=$1 MNERF_GEMM_SPLIT_MIN_K=$2 MNERF_GEMM_SPLIT_WGS=$3 timeout 120 python tools/exp/gemm_time.py 2>&1 | grep "weight grad" | cut -c1-75 >> gpurun_out/r4s_gemm.log
done
cat gpurun_out/r4s_gemm.log
