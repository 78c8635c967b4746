#!/bin/bash
# The bench lines a round keeps under profiles/: the default line (both halves of the metric, full T = 1000 call, full-range
# figures, CPU baselines), the other configs on request, per-launch-shape listings, and the self-spawning multi-rank path
# rehearsed with two gloo ranks on one GPU.
# Usage (GPU box, repo root): bash tools/bench_set.sh <tag>      -> gpurun_out/<tag>/...
set -o pipefail
tag=${1:-benchset}
out=gpurun_out/$tag
mkdir -p $out
run() {   # name, seconds, args...
  local name=$1 secs=$2; shift 2
  echo "== $name: python bench.py $*"
  timeout -k 10 $secs python bench.py "$@" > $out/$name.json 2> $out/$name.err || { tail -5 $out/$name.err; exit 1; }
  grep "ms/step\|full call\|images/s" $out/$name.err | tail -4
}
run bench_default 560
run bench_sample_c1 120 --workload sample_c1
run bench_train_color 200 --workload train_color --no-cpu-baseline
run bench_sample_64 300 --workload sample_64 --batch 1024 --no-cpu-baseline --no-full-call
run bench_sample_512 200 --workload sample --batch 512 --no-cpu-baseline
DDPM_BENCH_PER_SHAPE=1 run per_shape_train 120 --workload train --profile-stride 1 --no-cpu-baseline --no-full-call --no-full-range
DDPM_BENCH_PER_SHAPE=1 run per_shape_sample 120 --workload sample --profile-stride 1 --steps 4 --no-cpu-baseline --no-full-call --no-full-range
run dist_selfspawn_gloo_2rank 300 --gpus 2 --backend gloo --same-device --steps 3 --warmup 1 --no-cpu-baseline --no-full-call --no-full-range
echo done
