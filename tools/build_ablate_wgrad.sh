#!/bin/bash
# Diagnostic libraries of the split-bf16 weight-gradient kernel with parts of its step removed:
#   tools/build_ablate_wgrad.sh 1 2 4 ...   -> ablate_build/libwg_<mask>.so  (select with DDPM_LIB=...)
set -e
cd "$(dirname "$0")/.."
mkdir -p ablate_build
objs=$(ls diffusion-model_amd/csrc/build/*.o | grep -v conv_wgrad)
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -fno-gpu-rdc -munsafe-fp-atomics \
     -DDDPM_WG_ABLATE=$m -x hip -c diffusion-model_amd/csrc/conv_wgrad.hip -o ablate_build/wg_$m.o &
done
wait
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ablate_build/libwg_$m.so $objs ablate_build/wg_$m.o
done
ls -la ablate_build/libwg_*.so
