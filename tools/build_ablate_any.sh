#!/bin/bash
# Diagnostic libraries with parts of one kernel removed (wrong results, for timing only):
#   tools/build_ablate_any.sh attention DDPM_ATTN_ABLATE 1 2 4   -> ablate_build/lib_<source>_<mask>.so  (select with DDPM_LIB=...)
set -e
cd "$(dirname "$0")/.."
src=$1; macro=$2; shift 2
mkdir -p ablate_build
objs=$(ls diffusion-model_amd/csrc/build/*.o | grep -v "/$src\.")
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -fno-gpu-rdc -munsafe-fp-atomics \
     -D$macro=$m -x hip -c diffusion-model_amd/csrc/$src.hip -o ablate_build/${src}_$m.o &
done
wait
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ablate_build/lib_${src}_$m.so $objs ablate_build/${src}_$m.o
done
ls ablate_build/*.so
