#!/bin/bash
# A/B of the two split arithmetics (three bf16 planes, six products / fp16 pair, three products) on the 3x3 and 1x1
# convolution shapes of the sampling (512 x 1x32x32) and training (256 x 1x28x28) workloads.
set -o pipefail
for math in bf16x3 f16x2; do
  echo "== DDPM_CONV_MATH=$math"
  for s in "512 32 32 64 64" "512 32 32 128 64" "512 16 16 128 128" "512 16 16 256 128" "512 8 8 256 256" "512 8 8 512 256" "256 28 28 64 64" "256 14 14 128 128" "256 7 7 256 256"; do
    DDPM_CONV_MATH=$math timeout -k 10 60 python tools/microbench.py conv3x3 $s 20 || exit 1
  done
  for s in "512 32 32 64 80" "512 16 16 128 160" "512 8 8 256 320"; do
    DDPM_CONV_MATH=$math timeout -k 10 60 python tools/microbench.py conv1x1 $s 20 || exit 1
  done
done
