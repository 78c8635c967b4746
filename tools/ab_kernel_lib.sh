#!/bin/bash
# Per-kernel A/B of two builds of the library under rocprofv3 --kernel-trace: tools/ab_kernel_lib.sh <other.so> <microbench args...>
# prints the average duration of every kernel of the run for the shipped library and for the other one.
other=$1; shift
export TMPDIR=/tmp
for lib in shipped $other; do
  d=gpurun_out/abk_$(basename $lib .so)
  rm -rf $d
  if [ $lib = shipped ]; then unset DDPM_LIB; else export DDPM_LIB=$lib; fi
  timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $d -o r -- python tools/microbench.py "$@" > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
  echo "== $lib"
  python tools/rocpd_stats.py $(ls $d/*.db | head -1) | sed -n 4,5p | cut -c1-150
  rm -rf $d
done
