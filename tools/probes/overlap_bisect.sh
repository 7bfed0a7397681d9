mkdir -p gpurun_out/r06d
for parts in 0 1 2 4 6 7 0 7; do
  if [ $parts = 0 ]; then export EGR_TRAIN_OVERLAP=0; else export EGR_TRAIN_OVERLAP=1; fi
  EGR_TRAIN_OVERLAP_PARTS=$parts python tools/train_bench.py --batch 32 --graph --steps 10 > gpurun_out/r06d/tb_$parts.txt 2>&1
  echo "parts=$parts rc=$? $(grep 'ms/step, ' gpurun_out/r06d/tb_$parts.txt)"
done
