#!/bin/bash
# 3x3 weight gradient of layers with 64 channels on one side: two taps per 128 x 128 tile against one tap per tile.
set -o pipefail
for shape in "256 28 28 64 128" "256 28 28 128 64" "512 32 32 64 128" "512 32 32 128 64"; do
  for m in 1 0; do
    echo -n "pair=$m  "
    DDPM_WGRAD_PAIR=$m timeout -k 10 90 python tools/microbench.py wgrad $shape 30 2>/dev/null || exit 1
  done
done
