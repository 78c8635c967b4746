#!/bin/bash
# Attention forward + backward microbenchmarks on the layer shapes of the bench configurations.
set -o pipefail
for shape in "256 784 128" "256 784 64" "512 1024 128" "512 1024 64" "512 256 256" "256 196 256"; do
  timeout -k 10 90 python tools/microbench.py attn $shape 50 2>/dev/null || exit 1
done
