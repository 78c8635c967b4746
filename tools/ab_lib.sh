#!/bin/bash
# A/B of the built library against another build of it: tools/ab_lib.sh <other.so> <microbench args...>, e.g.
#   tools/ab_lib.sh ablate_build/lib_base_bwd.so attn 256 784 128 50
set -o pipefail
other=$1; shift
echo -n "this build: "; timeout -k 10 90 python tools/microbench.py "$@" 2>/dev/null | tr '\n' ';'; echo
echo -n "$other: "; DDPM_LIB=$PWD/$other timeout -k 10 90 python tools/microbench.py "$@" 2>/dev/null | tr '\n' ';'; echo
