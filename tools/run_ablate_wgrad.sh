#!/bin/bash
set -o pipefail
shape=${1:-"256 28 28 128 128"}
timeout -k 10 60 python tools/microbench.py wgrad $shape || exit 1
for f in ablate_build/libwg_*.so; do
  echo "== $f"
  DDPM_LIB=$PWD/$f timeout -k 10 60 python tools/microbench.py wgrad $shape || exit 1
done
