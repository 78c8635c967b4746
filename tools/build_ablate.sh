#!/bin/bash
# Diagnostic libraries of the halo conv kernel with parts of its loop removed:
#   tools/build_ablate.sh 1 2 4 ...   -> ablate_build/lib_<mask>.so  (select with DDPM_LIB=...)
set -e
cd "$(dirname "$0")/.."
mkdir -p ablate_build
objs=$(ls diffusion-model_amd/csrc/build/*.o | grep -v conv3x3_halo)
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -fno-gpu-rdc -munsafe-fp-atomics \
     -DDDPM_HALO_ABLATE=$m -x hip -c diffusion-model_amd/csrc/conv3x3_halo.hip -o ablate_build/halo_$m.o &
done
wait
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ablate_build/lib_$m.so $objs ablate_build/halo_$m.o
done
ls -la ablate_build/*.so
