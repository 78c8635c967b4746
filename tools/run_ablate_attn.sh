#!/bin/bash
# Runs every ablate_build/lib_attention_*.so on two attention shapes (forward line only).
set -o pipefail
for shape in "512 1024 128" "512 1024 64"; do
  echo "#### $shape"
  timeout -k 10 60 python tools/microbench.py attn $shape 50 2>/dev/null | grep fwd || exit 1
  for f in ablate_build/lib_attention_*.so; do
    echo -n "== $f  "
    DDPM_LIB=$PWD/$f timeout -k 10 60 python tools/microbench.py attn $shape 50 2>/dev/null | grep fwd || exit 1
  done
done
