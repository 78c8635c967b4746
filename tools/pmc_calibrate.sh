#!/bin/bash
# Calibration of FETCH_SIZE on this chip against known byte counts, per access pattern:
#   reverse step (16 B per lane, fully coalesced streams; reads 12 B per element),
#   3x3 halo conv with ONE column tile (64-byte pieces of 256-byte pixel rows; reads the input once plus halo re-reads),
#   1x1 conv with resident weights (two 16-byte loads per lane, 32 consecutive bytes of a pixel row).
# All tensors are larger than the 256 MiB on-die cache, so every read reaches the memory-side counters.
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/pmc_cal
rm -rf $out; mkdir -p $out
run() {  # tag, microbench args...
  tag=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $out/$tag -o p -- python tools/microbench.py "$@" > $out/$tag.log 2>&1 || { tail -5 $out/$tag.log; exit 1; }
  python - "$out/$tag" "$tag" <<'PY'
import glob, sys
sys.path.insert(0, "tools")
from pmc_traffic import per_kernel
f = per_kernel(glob.glob(sys.argv[1] + "/*.db")[0], "FETCH_SIZE")
for k in sorted(f, key=lambda k: -f[k][0])[:3]:
    print(f"{sys.argv[2]:10s} raw FETCH_SIZE {f[k][0] * 1024 / f[k][1] / 1e6:9.1f} MB per dispatch ({f[k][1]} dispatches)  {k[:90]}")
PY
  rm -f $out/$tag/*.db
}
run step step 67108864 5 || exit 1
run conv3x3 conv3x3 2048 32 32 64 64 3 || exit 1
run conv1x1 conv1x1 2048 32 32 64 80 3 || exit 1
# the same input through 2 and 4 column tiles: do the workgroups that share rows find them in their XCD's L2?
run conv3x3n2 conv3x3 2048 32 32 64 128 3 || exit 1
run conv3x3n4 conv3x3 2048 32 32 64 256 3 || exit 1
