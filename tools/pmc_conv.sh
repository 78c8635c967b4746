#!/bin/bash
# Hardware counters of a microbenchmark kernel (one pass per counter set): tools/pmc_conv.sh "<shape>"; KIND=wgrad PAT=conv_wgrad_x3 selects another kernel.
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/pmc_conv
mkdir -p $out
shape=${1:-"256 28 28 128 128"}
timeout -s KILL 90 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -d $out/a -o a -- python tools/microbench.py ${KIND:-conv3x3} $shape 5 > $out/a.log 2>&1 || { tail -5 $out/a.log; exit 1; }
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d $out/b -o b -- python tools/microbench.py ${KIND:-conv3x3} $shape 5 > $out/b.log 2>&1 || { tail -5 $out/b.log; exit 1; }
timeout -s KILL 90 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $out/c -o c -- python tools/microbench.py ${KIND:-conv3x3} $shape 5 > $out/c.log 2>&1 || { tail -5 $out/c.log; exit 1; }
for f in $out/a/*.db $out/c/*.db $out/b/*.db; do python tools/pmc_summary.py $f | grep -A12 "${PAT:-conv3x3_halo\|conv_gemm}"; done
rm -f $out/*/*.db
