mkdir -p gpurun_out/r06d
for blk in 256 248 240 224; do
  EGR_CONV_TAPX_BLOCKS=$blk EGR_CONV_PW_BLOCKS=$blk python tools/train_bench.py --batch 32 --graph --steps 10 > gpurun_out/r06d/tb_b$blk.txt 2>&1
  echo "blocks=$blk rc=$? $(grep 'ms/step, ' gpurun_out/r06d/tb_b$blk.txt)"
done
