#!/bin/bash
# 1x1 convolutions on the resident-weights / slab kernels: the shipped library against ablate_build/lib_c1r_old.so.
set -o pipefail
cd "$(dirname "$0")/.."
for shape in "512 32 32 128 160" "512 32 32 64 80" "512 16 16 256 320" "256 28 28 128 160" "256 28 28 64 80" "256 14 14 256 320" "1024 64 64 64 80"; do
  for lib in "" ablate_build/lib_c1r_old.so; do
    echo -n "${lib:-shipped}  "
    if [ -n "$lib" ]; then export DDPM_LIB=$PWD/$lib; else unset DDPM_LIB; fi
    timeout -k 10 60 python tools/microbench.py conv1x1 $shape 2>/dev/null || exit 1
  done
done
