#!/bin/bash
# Whole-step A/B of the built library against ablate_build/lib_base.so, alternating, twice each:
#   bash tools/ab_step.sh <train|sample>
set -o pipefail
wl=${1:-sample}
for i in 1 2; do
  for lib in this base; do
    if [ $lib = base ]; then export DDPM_LIB=$PWD/ablate_build/lib_base.so; else unset DDPM_LIB; fi
    timeout -k 10 200 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; h=[v for n,v in k.items() if 'halo' in n][0]
print('$lib', d['ms_per_step'], 'halo avg ms', h['avg_ms'], 'tflops', h['tflops'])" || exit 1
  done
done
