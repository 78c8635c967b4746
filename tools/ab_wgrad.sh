#!/bin/bash
# Weight-gradient microbenchmark on the layer shapes of the training bench (run on the GPU box).
set -o pipefail
for shape in "256 28 28 128 128" "256 28 28 64 128" "256 14 14 256 256" "256 14 14 128 256" "256 7 7 512 512" "256 3 3 512 512"; do
  timeout -k 10 60 python tools/microbench.py wgrad $shape || exit 1
done
