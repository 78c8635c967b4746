#!/bin/bash
# A/B of the two split arithmetics on the 3x3 weight-gradient shapes of the training workload (256 x 1x28x28).
set -o pipefail
for math in bf16x3 f16x2; do
  echo "== DDPM_CONV_MATH=$math"
  for s in "256 28 28 64 128" "256 28 28 128 128" "256 28 28 256 128" "256 14 14 128 256" "256 14 14 256 256" "256 14 14 512 256" "256 7 7 256 512" "256 7 7 512 512"; do
    DDPM_CONV_MATH=$math timeout -k 10 60 python tools/microbench.py wgrad $s 20 2>/dev/null || exit 1
  done
done
