#!/bin/bash
# bench.py (inference only) under different size thresholds of the split-bf16 launches: tools/x6_threshold_sweep.sh
mkdir -p gpurun_out
for cfg in "8192 4e9" "4096 2e9" "4096 1e9" "2048 1e9" "2048 5e8" "8192 4e9"; do
  set -- $cfg
  EGR_X6_MIN_ROWS=$1 EGR_X6_MIN_FLOPS=$2 python bench.py --no-train --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernel_ms']
print('rows $1 flops $2:', d['value'], 'f/s', d['ms_per_step'], 'ms | x6', k.get('egr_conv2d_nhwc_f32[bf16x3]'), 'f32', k.get('egr_conv2d_nhwc_f32'), 'launches x6', d['roofline']['launches_per_step'], 'frac', d['roofline']['frac'])"
done
