#!/bin/bash
# batch 1 .. 16 latency against the size rule that sends a launch to the split kernels (rows per group x groups, FLOPs)
for f in 5e8 2e8 5e7; do for r in 4096 2048 1024 512; do
echo "=== flops $f rows $r"; EGR_X6_MIN_FLOPS=$f EGR_X6_MIN_ROWS=$r timeout -k 10 120 python tools/latency_small.py 2>&1 | grep "B=" | tr '\n' ' '; echo; done; done
