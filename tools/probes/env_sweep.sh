#!/bin/bash
# bench.py (inference only) under different values of one environment knob:  tools/env_sweep.sh VAR v1 v2 ...
var=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
  env $var=$v python bench.py --no-train --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernel_ms']
print('$var=$v:', d['value'], 'f/s', d['ms_per_step'], 'ms | x6', k.get('egr_conv2d_nhwc_f32[bf16x3]'), 'f32', k.get('egr_conv2d_nhwc_f32'), 'frac', d['roofline']['frac'])"
done
