#!/bin/bash
# Hardware counters of the 1x1 conv microbenchmark (one pass per counter set).
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/pmc_c1
mkdir -p $out
shape=${1:-"512 32 32 128 160"}
timeout -s KILL 90 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -d $out/a -o a -- python tools/microbench.py conv1x1 $shape 5 > $out/a.log 2>&1 || { tail -5 $out/a.log; exit 1; }
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD -d $out/b -o b -- python tools/microbench.py conv1x1 $shape 5 > $out/b.log 2>&1 || { tail -5 $out/b.log; exit 1; }
timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES SQ_INSTS_SMEM -d $out/d -o d -- python tools/microbench.py conv1x1 $shape 5 > $out/d.log 2>&1 || { tail -5 $out/d.log; exit 1; }
timeout -s KILL 90 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $out/c -o c -- python tools/microbench.py conv1x1 $shape 5 > $out/c.log 2>&1 || { tail -5 $out/c.log; exit 1; }
for f in $out/a/*.db $out/b/*.db $out/d/*.db $out/c/*.db; do python tools/pmc_summary.py $f | grep -A12 "conv1x1_resident"; done
