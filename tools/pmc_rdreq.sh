#!/bin/bash
# Memory-side read requests of the L2 by size class for one microbenchmark (to interpret FETCH_SIZE).
# Usage: bash tools/pmc_rdreq.sh <tag> <microbench args...>
set -o pipefail
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/pmc_rdreq_$tag
rm -rf $out; mkdir -p $out
timeout -s KILL 60 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum -d $out/a -o p -- python tools/microbench.py "$@" > $out/a.log 2>&1 || { tail -8 $out/a.log; exit 1; }
python tools/pmc_summary.py $out/a/*.db | head -12
rm -f $out/a/*.db
