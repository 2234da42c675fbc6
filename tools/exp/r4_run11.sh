#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 120 tools/exp/ubench/dma_rate > gpurun_out/r4k_dma_rate.log 2>&1
cat gpurun_out/r4k_dma_rate.log
