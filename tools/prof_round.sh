#!/bin/bash
# Kernel-time profiles of both halves of the bench (rocprofv3 --kernel-trace --stats), one run per workload.
# The training run keeps the backward on ONE stream so that every kernel runs alone, as the launches that
# bench.py brackets with HIP events do, and the two average durations can be compared.
# Usage (GPU box, repo root): bash tools/prof_round.sh <tag>
set -o pipefail
tag=${1:-round}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
for wl in train sample; do
  if [ $wl = train ]; then export DDPM_SIDE_STREAM=0; else unset DDPM_SIDE_STREAM; fi
  timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $out/prof_$wl -o r -- python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-full-call --no-full-range --settle 0.5 > $out/prof_$wl.log 2>&1 || { tail -20 $out/prof_$wl.log; exit 1; }
  db=$(ls $out/prof_$wl/*.db | head -1)
  python tools/rocpd_stats.py $db > $out/$wl.stats.txt || exit 1
  rm -f $out/prof_$wl/*.db
  head -12 $out/$wl.stats.txt | cut -c1-160
done
