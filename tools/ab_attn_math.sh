#!/bin/bash
# A/B of the two split arithmetics on the attention shapes of the sampling (512 x 32x32), training (256 x 28x28) and
# 64x64 workloads: forward and backward.
set -o pipefail
for math in bf16x3 f16x2; do
  echo "== DDPM_CONV_MATH=$math"
  for s in "512 1024 64" "512 1024 128" "512 256 256" "256 784 64" "256 784 128" "256 196 256" "64 4096 64" "64 4096 128"; do
    DDPM_CONV_MATH=$math timeout -k 10 90 python tools/microbench.py attn $s 10 2>/dev/null || exit 1
  done
done
