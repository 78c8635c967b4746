#!/bin/bash
# Attention forward, inference and training calls: the shipped library against ablate_build/lib_attn_old.so (the running
# maximum following every tile) on the sampling, training and 64x64 shapes.
set -o pipefail
cd "$(dirname "$0")/.."
for s in "512 1024 64" "512 1024 128" "512 256 256" "256 784 128" "64 4096 128"; do
  for lib in "" ablate_build/lib_attn_old.so; do
    echo "== ${lib:-shipped}"
    if [ -n "$lib" ]; then export DDPM_LIB=$PWD/$lib; else unset DDPM_LIB; fi
    timeout -k 10 90 python tools/microbench.py attn $s 10 2>/dev/null | grep fwd || exit 1
  done
done
