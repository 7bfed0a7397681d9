#!/bin/bash
# usage: tools/run_env_variants.sh "VAR=a VAR=b ..." [conv_micro args]  (each run: env VAR=value python tools/conv_micro.py --x6 ...)
sets="$1"; shift
mkdir -p gpurun_out
for e in $sets; do
  echo "=== $e" | tee gpurun_out/envvar_$e.txt
  env $e timeout -k 10 300 python tools/conv_micro.py --x6 "$@" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/envvar_$e.txt || exit 1
done
