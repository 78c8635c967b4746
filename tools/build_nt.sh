#!/bin/bash
# Diagnostic library whose convolution epilogues store with another cache policy (DDPM_STORE_AUX: 2 = non-temporal):
#   tools/build_nt.sh [aux] [files...]   -> ablate_build/lib_nt.so  (select with DDPM_LIB=...)
set -e
cd "$(dirname "$0")/.."
aux=${1:-2}; shift || true
files=${@:-conv3x3_halo conv_gemm conv1x1_resident}
mkdir -p ablate_build
pat=$(echo $files | sed 's/ /\\|/g')
objs=$(ls diffusion-model_amd/csrc/build/*.o | grep -v "/\($pat\)\.o")
for f in $files; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -fno-gpu-rdc -munsafe-fp-atomics \
     -DDDPM_STORE_AUX=$aux -x hip -c diffusion-model_amd/csrc/$f.hip -o ablate_build/${f}_nt.o &
done
wait
ntobjs=$(for f in $files; do echo ablate_build/${f}_nt.o; done)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ablate_build/lib_nt.so $objs $ntobjs
ls -la ablate_build/lib_nt.so
