#!/bin/bash
# 1x1 convolution shapes of the training bench: q|k|v projections and their data gradients.
set -o pipefail
for shape in "256 28 28 64 80" "256 28 28 80 64" "256 14 14 128 160" "256 14 14 160 128" "256 28 28 128 160" "256 28 28 160 128" "256 7 7 256 320" "256 7 7 320 256" "256 3 3 512 640"; do
  timeout -k 10 60 python tools/microbench.py conv1x1 $shape || exit 1
done
