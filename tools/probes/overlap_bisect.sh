mkdir -p gpurun_out/r06h
for parts in 7 15 7 15; do
  EGR_TRAIN_OVERLAP_PARTS=$parts python tools/train_bench.py --batch 32 --graph --steps 10 > gpurun_out/r06h/tb_p$parts.txt 2>&1
  echo "parts=$parts rc=$? $(grep 'ms/step, ' gpurun_out/r06h/tb_p$parts.txt)"
done
