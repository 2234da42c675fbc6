#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the decoder kernel for the library in $MNERF_LIB (separate --pmc passes):  pmc_decoder_traffic.sh TAG
TAG=${1:-t}; R=${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/pt_$TAG
for P in c d; do
  case $P in c) C="FETCH_SIZE TCC_HIT_sum";; d) C="WRITE_SIZE TCC_MISS_sum";; esac
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $C -d /tmp/pt_$TAG/$P -- python $R/tools/prof_render.py 1 > /dev/null 2>&1
done
python $R/tools/pmc_summary.py /tmp/pt_$TAG 2>&1 | grep -A6 "decoder_pp"
