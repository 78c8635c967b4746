#!/bin/bash
# A/B of the one-workgroup-per-sample attention against the tiled kernels at the bottleneck shapes.
set -o pipefail
for shape in "256 49 512" "512 64 512" "256 9 512"; do
  echo "== $shape tiled"; DDPM_ATTN_SMALL=0 timeout -k 10 100 python tools/microbench.py attn $shape 20 || exit 1
  for kib in 80 160; do
    echo "== $shape small, $kib KiB"; DDPM_ATTN_SMALL_LDS=$kib timeout -k 10 100 python tools/microbench.py attn $shape 20 || exit 1
  done
done
