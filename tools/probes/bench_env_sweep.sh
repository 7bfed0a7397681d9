#!/bin/bash
# usage: tools/persist_sweep.sh  -- bench.py (inference leg only) under a few environment settings, one line each
for e in "EGR_X6_MIN_ROWS=4096" "EGR_X6_MIN_ROWS=2048" "EGR_X6_MIN_ROWS=1024" "EGR_X6_MIN_ROWS=512 EGR_X6_MIN_FLOPS=2e8"; do echo "=== $e"; env $e timeout -k 10 200 python bench.py --no-train --no-configs --no-cpu-baseline 2>&1 | tail -1 | cut -c1-175; done
