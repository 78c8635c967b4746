#!/bin/bash
# A/B of one environment switch on one box, alternating: tools/ab_env.sh VAR=value <workload> [rounds] [extra bench args]
# prints ms/step without and with the setting per round.
kv=$1; wl=${2:-train}; rounds=${3:-3}; shift 3 || true
for i in $(seq 1 $rounds); do
  a=$(timeout -k 10 200 python bench.py --workload $wl --steps 20 --no-cpu-baseline --no-full-call --no-full-range "$@" 2>&1 >/dev/null | grep -o "[0-9.]* ms/step" | tail -1) || exit 1
  b=$(env $kv timeout -k 10 200 python bench.py --workload $wl --steps 20 --no-cpu-baseline --no-full-call --no-full-range "$@" 2>&1 >/dev/null | grep -o "[0-9.]* ms/step" | tail -1) || exit 1
  echo "$wl round $i: default $a   $kv $b"
done
