set -o pipefail
for shape in "256 28 28 128 160" "256 28 28 64 80" "256 28 28 160 128" "256 28 28 80 64" "512 32 32 128 160" "512 32 32 64 80"; do
  timeout -k 10 60 python tools/microbench.py conv1x1 $shape 2>/dev/null || exit 1
  DDPM_CONV1X1_RESIDENT=0 timeout -k 10 60 python tools/microbench.py conv1x1 $shape 2>/dev/null || exit 1
done
