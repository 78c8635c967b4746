#!/bin/bash
# Attention forward: 32 queries x 128 channels per wave (default for C >= 128) against the 64 x 64 configuration.
set -o pipefail
for shape in "512 1024 128" "256 784 128" "512 256 256" "256 196 256" "64 4096 128"; do
  timeout -k 10 90 python tools/microbench.py attn $shape 2>/dev/null | grep fwd || exit 1
  DDPM_ATTN_WIDE=0 timeout -k 10 90 python tools/microbench.py attn $shape 2>/dev/null | grep fwd || exit 1
done
