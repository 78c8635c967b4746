#!/bin/bash
# Full GPU checkpoint: tests, both benches, rocprof kernel stats for both workloads.
# Usage (on the GPU box, from the repo root): bash tools/checkpoint_gpu.sh <tag>
set -o pipefail
tag=${1:-ckpt}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $out/tests.log 2>&1 || { tail -20 $out/tests.log; exit 1; }
tail -2 $out/tests.log
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > $out/bench_train.json 2> $out/bench_train.err || { tail -20 $out/bench_train.err; exit 1; }
timeout -k 10 200 python bench.py --workload sample --steps 20 --warmup 5 > $out/bench_sample.json 2> $out/bench_sample.err || { tail -20 $out/bench_sample.err; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/prof_train -o r -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/prof_train.log 2>&1 || { tail -20 $out/prof_train.log; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/prof_sample -o r -- python bench.py --workload sample --steps 5 --warmup 2 --no-cpu-baseline > $out/prof_sample.log 2>&1 || { tail -20 $out/prof_sample.log; exit 1; }
for w in train sample; do
  db=$(ls $out/prof_$w/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python tools/rocpd_stats.py $db > $out/${w}.stats.txt
done
python - <<PY
import json
for w in ("train","sample"):
    j=json.load(open("$out/bench_%s.json"%w))
    print(w, j["value"], j["unit"], j["ms_per_step"], "ms/step", j["roofline"]["kernel"], j["roofline"]["achieved"], j.get("cpu_baseline",{}).get("value"))
PY
