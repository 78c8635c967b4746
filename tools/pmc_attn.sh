#!/bin/bash
# Hardware counters of the attention microbenchmark (one pass per counter set).
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/pmc_attn
mkdir -p $out
shape=${1:-"512 1024 128"}
timeout -s KILL 90 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -d $out/a -o a -- python tools/microbench.py attn $shape 3 > $out/a.log 2>&1 || { tail -5 $out/a.log; exit 1; }
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAVES -d $out/b -o b -- python tools/microbench.py attn $shape 3 > $out/b.log 2>&1 || { tail -5 $out/b.log; exit 1; }
timeout -s KILL 90 rocprofv3 --pmc GRBM_GUI_ACTIVE -d $out/c -o c -- python tools/microbench.py attn $shape 3 > $out/c.log 2>&1 || { tail -5 $out/c.log; exit 1; }
for f in $out/a/*.db $out/b/*.db $out/c/*.db; do python tools/pmc_summary.py $f | grep -A9 "attention_fwd"; done
rm -f $out/*/*.db
