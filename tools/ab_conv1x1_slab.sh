#!/bin/bash
# 1x1 convolutions whose weights do not fit LDS: column tiles + K slabs (default) against the generic GEMM kernel.
set -o pipefail
for shape in "512 8 8 512 640" "512 16 16 256 320" "512 8 8 512 512" "512 16 16 256 256" "256 14 14 256 320" "256 14 14 256 256" "1024 16 16 512 640"; do
  for m in 1 0; do
    echo -n "slab=$m  "
    DDPM_CONV1X1_RESIDENT=$m timeout -k 10 90 python tools/microbench.py conv1x1 $shape 50 2>/dev/null || exit 1
  done
done
