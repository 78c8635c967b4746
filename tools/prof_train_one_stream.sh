#!/bin/bash
# rocprofv3 kernel stats of the training bench with the backward on ONE stream: every kernel then runs alone,
# as the launches that bench.py brackets with HIP events do, so the two average durations can be compared.
set -o pipefail
out=gpurun_out/${1:-one_stream}
mkdir -p $out
export TMPDIR=/tmp DDPM_SIDE_STREAM=0
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_train.json 2> $out/bench_train.err || { tail -20 $out/bench_train.err; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/prof_train -o r -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/prof_train.log 2>&1 || { tail -20 $out/prof_train.log; exit 1; }
db=$(ls $out/prof_train/*.db | head -1)
python tools/rocpd_stats.py $db > $out/train.stats.txt
rm -f $out/prof_train/*.db
head -8 $out/train.stats.txt | cut -c1-150
python -c "import json; j=json.load(open('$out/bench_train.json')); print(j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['achieved'])"
