#!/bin/bash
# Counters of one microbenchmark kernel. Usage: bash tools/pmc_generic.sh <kind> "<shape>" <kernel-name-pattern>
set -o pipefail
export TMPDIR=/tmp
kind=$1; shape=$2; pat=$3
out=gpurun_out/pmc_generic
rm -rf $out; mkdir -p $out
timeout -s KILL 90 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES -d $out/a -o a -- python tools/microbench.py $kind $shape 5 > $out/a.log 2>&1 || { tail -5 $out/a.log; exit 1; }
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM -d $out/b -o b -- python tools/microbench.py $kind $shape 5 > $out/b.log 2>&1 || { tail -5 $out/b.log; exit 1; }
timeout -s KILL 90 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -d $out/c -o c -- python tools/microbench.py $kind $shape 5 > $out/c.log 2>&1 || { tail -5 $out/c.log; exit 1; }
timeout -s KILL 90 rocprofv3 --pmc GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum -d $out/d -o d -- python tools/microbench.py $kind $shape 5 > $out/d.log 2>&1 || { tail -5 $out/d.log; exit 1; }
for f in $out/a/*.db $out/b/*.db $out/c/*.db $out/d/*.db; do python tools/pmc_summary.py $f | grep -A9 "$pat"; done
