#!/bin/bash
# round 5, call 3: A/B of two decoder builds against the shipped one (same box): SP=256 instance of decoder_kernel with one workgroup per CU
# (mb1), weight requests with one stream base + vector offset (voff)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r5_3; mkdir -p $O
L=$PWD/matchnerf_amd
for rep in 1 2; do
  for v in "" voff; do
    lib=$L/libmnerf_hip${v:+_$v}.so
    for c in c2 s128; do MNERF_LIB=$lib timeout 300 python tools/exp/frame_time.py $c 6 2>&1 | tail -1 | sed "s/^/[${v:-base}] /" | tee -a $O/frames.log; done
  done
done
for v in "" mb1; do
  lib=$L/libmnerf_hip${v:+_$v}.so
  MNERF_LIB=$lib timeout 300 python tools/exp/frame_time.py s256 3 2>&1 | tail -1 | sed "s/^/[${v:-base}] /" | tee -a $O/frames.log
done
