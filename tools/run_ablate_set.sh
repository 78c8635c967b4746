#!/bin/bash
# Runs every library in ablate_build/ (tools/build_ablate.sh) on three layer shapes of the halo kernel.
set -o pipefail
for shape in "512 32 32 64 64" "256 28 28 128 128" "512 16 16 256 256"; do
  echo "#### $shape"
  timeout -k 10 60 python tools/microbench.py conv3x3 $shape 2>/dev/null || exit 1
  for f in ablate_build/lib_*.so; do
    echo -n "== $f  "
    DDPM_LIB=$PWD/$f timeout -k 10 60 python tools/microbench.py conv3x3 $shape 2>/dev/null || exit 1
  done
done
