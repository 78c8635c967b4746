#!/bin/bash
# 3x3 conv shapes of the third-iteration UNet at the sampling batch (512 per GPU), 1x32x32 input.
set -o pipefail
for s in "512 32 32 64 64" "512 32 32 128 64" "512 16 16 64 128" "512 16 16 128 128" "512 16 16 256 128" "512 8 8 128 256" "512 8 8 256 256" "512 8 8 256 512" "512 8 8 512 512" "512 16 16 512 256" "512 16 16 256 256" "512 32 32 256 128" "512 32 32 128 128"; do
  timeout -k 10 60 python tools/microbench.py conv3x3 $s 10 || exit 1
done
