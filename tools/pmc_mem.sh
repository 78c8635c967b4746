#!/bin/bash
# L2 / memory-side counters of a microbenchmark kernel. Usage: bash tools/pmc_mem.sh <kind> "<shape>" <kernel-name-pattern>
set -o pipefail
export TMPDIR=/tmp
kind=${1:-wgrad}
shape=${2:-"256 28 28 128 128"}
pat=${3:-conv_wgrad}
out=gpurun_out/pmc_mem
mkdir -p $out
timeout -s KILL 90 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $out/a -o a -- python tools/microbench.py $kind $shape 5 > $out/a.log 2>&1 || { tail -5 $out/a.log; exit 1; }
timeout -s KILL 90 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum -d $out/b -o b -- python tools/microbench.py $kind $shape 5 > $out/b.log 2>&1 || { tail -5 $out/b.log; exit 1; }
for f in $out/a/*.db $out/b/*.db; do python tools/pmc_summary.py $f | grep -A6 "$pat"; done
