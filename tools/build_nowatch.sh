#!/bin/bash
# Diagnostic library whose fp16-pair kernels do not watch their operands' range (prices the watch): ablate_build/lib_nowatch.so
set -e
cd "$(dirname "$0")/.."
mkdir -p ablate_build
files="conv3x3_halo conv_gemm conv1x1_resident conv_wgrad attention attention_bwd"
pat=$(echo $files | sed 's/ /\\|/g')
objs=$(ls diffusion-model_amd/csrc/build/*.o | grep -v "/\($pat\)\.o")
for f in $files; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -fno-gpu-rdc -munsafe-fp-atomics \
     -DDDPM_NO_WATCH=1 -x hip -c diffusion-model_amd/csrc/$f.hip -o ablate_build/${f}_nowatch.o 2>/dev/null &
done
wait
nw=$(for f in $files; do echo ablate_build/${f}_nowatch.o; done)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ablate_build/lib_nowatch.so $objs $nw
ls -la ablate_build/lib_nowatch.so
