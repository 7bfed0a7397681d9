#!/bin/bash
# Same-box A/B of the two-stream training step (DESIGN.md 9 "Round 6"): EGR_TRAIN_OVERLAP_PARTS bits 1 = forward branches, 2 = the detached
# heads' reverse pass, 4 = the refiners' reverse pass; 0 = one stream.  Also used for EGR_TRAIN_STEM_FUSED=0/1 and EGR_CONV_TAPX_BLOCKS.
#   bash tools/probes/train_overlap_ab.sh            (on the GPU box, from the repository root)
mkdir -p gpurun_out/train_ab
for parts in 0 7 0 7; do
  if [ $parts = 0 ]; then export EGR_TRAIN_OVERLAP=0; else export EGR_TRAIN_OVERLAP=1; fi
  EGR_TRAIN_OVERLAP_PARTS=$parts python tools/train_bench.py --batch 32 --graph --steps 10 > gpurun_out/train_ab/tb_$parts.txt 2>&1
  echo "parts=$parts rc=$? $(grep 'ms/step, ' gpurun_out/train_ab/tb_$parts.txt)"
done
