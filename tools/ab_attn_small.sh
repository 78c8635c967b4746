#!/bin/bash
# Small-map attention (one workgroup per sample) on the bottleneck shapes: the shipped library against ablate_build/lib_as_old.so.
set -o pipefail
cd "$(dirname "$0")/.."
for shape in "256 49 512" "256 9 512" "512 16 512" "512 64 512" "256 49 1024"; do
  for lib in "" ablate_build/lib_as_old.so; do
    echo "== ${lib:-shipped}"
    if [ -n "$lib" ]; then export DDPM_LIB=$PWD/$lib; else unset DDPM_LIB; fi
    timeout -k 10 60 python tools/microbench.py attn $shape 20 2>/dev/null || exit 1
  done
done
