#!/bin/bash
for f in 4e9 2e9 1e9 5e8 2e8; do for r in 8192 4096 2048; do
echo "=== flops $f rows $r"; EGR_X6_MIN_FLOPS=$f EGR_X6_MIN_ROWS=$r timeout -k 10 120 python tools/latency_small.py 2>&1 | grep "B=" | tr '\n' ' '; echo; done; done
