#!/bin/bash
# like run_variants.sh but conv_micro arguments are passed verbatim (no implied --x6)
names="$1"; shift
mkdir -p gpurun_out
for n in $names; do
  lib=egorear_amd/csrc/libegorear_hip_$n.so
  [ "$n" = "main" ] && lib=egorear_amd/csrc/libegorear_hip.so
  echo "=== $n" | tee gpurun_out/variants_$n.txt
  EGR_LIB=$PWD/$lib timeout -k 10 300 python tools/conv_micro.py "$@" 2>&1 | tee -a gpurun_out/variants_$n.txt || exit 1
done
