#!/bin/bash
# Kernel-time profile of the training bench only. Usage (GPU box, repo root): bash tools/prof_train.sh <tag> [pattern]
set -o pipefail
tag=${1:-p}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/prof_train -o r -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/prof_train.log 2>&1 || { tail -20 $out/prof_train.log; exit 1; }
db=$(ls $out/prof_train/*.db | head -1)
python tools/rocpd_stats.py $db > $out/train.stats.txt
head -3 $out/train.stats.txt
grep -E "${2:-bn_}" $out/train.stats.txt | cut -c1-150
