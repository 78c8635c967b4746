#!/bin/bash
# L1 <-> L2 traffic of the 1x1 convolution microbenchmark: requests from the CUs' L1 caches to L2 against cache accesses.
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/pmc_conv1x1
mkdir -p $out
shape=${1:-"512 32 32 128 160"}
timeout -s KILL 90 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum -d $out/a -o a -- python tools/microbench.py conv1x1 $shape 5 > $out/a.log 2>&1 || { tail -5 $out/a.log; exit 1; }
timeout -s KILL 90 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum -d $out/b -o b -- python tools/microbench.py conv1x1 $shape 5 > $out/b.log 2>&1 || { tail -5 $out/b.log; exit 1; }
for f in $out/a/*.db $out/b/*.db; do python tools/pmc_summary.py $f | grep -A8 "conv1x1\|conv_gemm"; done
rm -f $out/*/*.db
