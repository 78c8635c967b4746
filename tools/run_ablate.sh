#!/bin/bash
set -o pipefail
shape=${1:-"256 28 28 128 128"}
timeout -k 10 60 python tools/microbench.py conv3x3 $shape || exit 1
for f in ablate_build/lib_*.so; do
  echo "== $f"
  DDPM_LIB=$PWD/$f timeout -k 10 60 python tools/microbench.py conv3x3 $shape || exit 1
done
