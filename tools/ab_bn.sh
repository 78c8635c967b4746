#!/bin/bash
# BatchNorm kernels at the layer shapes of the training workload (256 x 1x28x28, base 64) and of a colour batch.
set -e
cd "$(dirname "$0")/.."
for shape in "256 28 28 64" "256 14 14 128" "256 7 7 256" "256 3 3 512" "64 64 64 64"; do
    timeout -k 10 120 python tools/microbench.py bn $shape 30
done
