#!/bin/bash
# Runs the 1x1 resident-weights kernel and its diagnostic builds (ablate_build/lib_c1r_*.so, conv1x1_resident.hip compiled with
# -DDDPM_C1R_ABLATE=<mask>: 1 no output stores, 2 activations loaded once, 4 no matrix instructions) on the q|k|v projection shapes.
set -o pipefail
cd "$(dirname "$0")/.."
for shape in "512 32 32 128 160" "512 32 32 64 80" "256 28 28 128 160" "256 28 28 64 80" "1024 64 64 64 80"; do
  echo "#### $shape"
  timeout -k 10 60 python tools/microbench.py conv1x1 $shape 2>/dev/null || exit 1
  for f in ablate_build/lib_c1r_*.so; do
    echo -n "== $f  "
    DDPM_LIB=$PWD/$f timeout -k 10 60 python tools/microbench.py conv1x1 $shape 2>/dev/null || exit 1
  done
done
