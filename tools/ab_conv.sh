#!/bin/bash
# A/B of the two conv arithmetic paths on a few layer shapes (run on the GPU box).
set -o pipefail
for shape in "256 28 28 128 128" "256 28 28 64 64" "256 14 14 256 256" "256 7 7 512 512" "512 32 32 128 128" "512 16 16 256 256"; do
  timeout -k 10 60 python tools/microbench.py conv3x3 $shape || exit 1
  DDPM_CONV_MATH=fp32 timeout -k 10 60 python tools/microbench.py conv3x3 $shape || exit 1
done
