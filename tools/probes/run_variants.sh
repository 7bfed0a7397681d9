#!/bin/bash
# usage: tools/run_variants.sh "name1 name2 ..." [conv_micro args]   -> gpurun_out/variants_<name>.txt
names="$1"; shift
mkdir -p gpurun_out
for n in $names; do
  lib=egorear_amd/csrc/libegorear_hip_$n.so
  [ "$n" = "main" ] && lib=egorear_amd/csrc/libegorear_hip.so
  echo "=== $n" | tee gpurun_out/variants_$n.txt
  EGR_LIB=$PWD/$lib timeout -k 10 300 python tools/conv_micro.py --x6 "$@" 2>&1 | tee -a gpurun_out/variants_$n.txt || exit 1
done
