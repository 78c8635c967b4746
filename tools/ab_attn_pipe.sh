#!/bin/bash
# Attention forward: software-pipelined kernel against the phase-by-phase kernel.
set -o pipefail
for shape in "512 1024 128" "256 784 128" "512 256 256" "256 196 256" "64 4096 128"; do
  for m in 0 1; do
    echo -n "pipe=$m  "
    DDPM_ATTN_PIPE=$m timeout -k 10 90 python tools/microbench.py attn $shape 50 2>/dev/null | grep fwd || exit 1
  done
done
