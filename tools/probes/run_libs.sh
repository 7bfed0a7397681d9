#!/bin/bash
# usage: tools/run_libs.sh "name1 name2 ..." <python script and args>   -- runs the script once per library variant (EGR_LIB)
names="$1"; shift
mkdir -p gpurun_out
for n in $names; do
  lib=egorear_amd/csrc/libegorear_hip_$n.so
  [ "$n" = "main" ] && lib=egorear_amd/csrc/libegorear_hip.so
  echo "=== $n" | tee gpurun_out/libs_$n.txt
  EGR_LIB=$PWD/$lib timeout -k 10 300 python "$@" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/libs_$n.txt || exit 1
done
