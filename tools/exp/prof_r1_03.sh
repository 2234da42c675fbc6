#!/bin/bash
# Round-1 final profile set: kernel trace + three PMC passes (separate runs, no sys/hip tracing).
cd /root/repo; export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r1_03; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof -o r1c -- python $R/tools/prof_render.py 3 > $O/trace.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/prof -name '*.db' | head -1) > $O/kernel_stats.md 2>&1
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
B="SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT"
for P in a b c d; do
  case $P in a) C="$A";; b) C="$B";; c) C="FETCH_SIZE TCC_HIT_sum TCP_TCC_READ_REQ_sum";; d) C="WRITE_SIZE TCC_MISS_sum TCC_EA0_RDREQ_sum";; esac
  timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc $C -d /tmp/pmc_$P -- python $R/tools/prof_render.py 1 > $O/pmc_$P.log 2>&1
done
python $R/tools/pmc_summary.py /tmp > $O/pmc_summary.txt 2>&1
cd $R; python bench.py > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.json
