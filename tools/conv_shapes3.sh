#!/bin/bash
# The three reference layer shapes of the halo kernel (GELU epilogue), one line each.
set -o pipefail
for shape in "512 32 32 64 64" "256 28 28 128 128" "512 16 16 256 256" "256 28 28 64 64" "256 7 7 512 512"; do
  timeout -k 10 60 python tools/microbench.py conv3x3 $shape 2>/dev/null || exit 1
done
