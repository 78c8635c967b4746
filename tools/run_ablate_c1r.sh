#!/bin/bash
# Runs every library in ablate_build/ on two shapes of the resident 1x1 kernel.
set -o pipefail
for shape in "256 28 28 128 160" "512 32 32 128 160" "512 32 32 64 80"; do
  timeout -k 10 60 python tools/microbench.py conv1x1 $shape 2>/dev/null || exit 1
  for f in ablate_build/lib_*.so; do echo -n "$f "; DDPM_LIB=$PWD/$f timeout -k 10 60 python tools/microbench.py conv1x1 $shape 2>/dev/null || exit 1; done
done
