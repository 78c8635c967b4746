#!/bin/bash
# Samples board power and clocks (rocm-smi) while one microbenchmark runs for a few seconds.
# Usage (GPU box, repo root): bash tools/power_probe.sh conv3x3 512 16 16 256 256 4000
set -o pipefail
timeout -k 10 120 python tools/microbench.py "$@" &
pid=$!
sleep 6
for i in 1 2 3 4 5 6; do
  kill -0 $pid 2>/dev/null || break
  timeout -k 5 20 rocm-smi --showpower --showclocks 2>&1 | grep -i "power\|sclk\|mclk\|fclk" | tr '\n' ';'
  echo
done
wait $pid
