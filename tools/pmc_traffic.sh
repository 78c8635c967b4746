#!/bin/bash
# Memory-side traffic (FETCH_SIZE, WRITE_SIZE; one counter pass each) per kernel over a short bench run.
# Usage (GPU box, repo root): bash tools/pmc_traffic.sh <train|sample> <tag>
set -o pipefail
export TMPDIR=/tmp
wl=${1:-train}
tag=${2:-traffic}
out=gpurun_out/$tag
rm -rf $out; mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 150 rocprofv3 --pmc $c -d $out/$c -o p -- python bench.py --workload $wl --steps 3 --warmup 1 --settle 0 --no-cpu-baseline --no-full-call --no-full-range > $out/$c.log 2>&1 || { tail -5 $out/$c.log; exit 1; }
done
python tools/pmc_traffic.py $out/FETCH_SIZE/*.db $out/WRITE_SIZE/*.db > $out/traffic_$wl.txt
rm -f $out/*/*.db      # the summaries are what is kept (gpurun copies back at most 64 MiB)
head -30 $out/traffic_$wl.txt
