#!/bin/bash
# The built library against ablate_build/lib_base.so on the halo kernel's layer shapes (sampling and training batches).
set -o pipefail
for shape in "512 32 32 64 64" "512 32 32 128 64" "512 16 16 128 128" "512 16 16 256 128" "512 8 8 256 256" "512 8 8 512 256" "256 28 28 64 64" "256 28 28 128 128" "256 14 14 256 256" "256 7 7 512 512"; do
  echo -n "this build: "; timeout -k 10 60 python tools/microbench.py conv3x3 $shape 30 2>/dev/null || exit 1
  echo -n "      base: "; DDPM_LIB=$PWD/ablate_build/lib_base.so timeout -k 10 60 python tools/microbench.py conv3x3 $shape 30 2>/dev/null || exit 1
done
