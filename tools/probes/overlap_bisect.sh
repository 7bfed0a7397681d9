mkdir -p gpurun_out/r06g
python -m pytest tests/test_gpu_train_kernels.py -x -q -m gpu -k "stem_batchnorm or batchnorm_train or maxpool" > gpurun_out/r06g/k.log 2>&1; tail -3 gpurun_out/r06g/k.log
for f in 0 1 0 1; do
  EGR_TRAIN_STEM_FUSED=$f python tools/train_bench.py --batch 32 --graph --steps 10 > gpurun_out/r06g/tb_f$f.txt 2>&1
  echo "stem_fused=$f rc=$? $(grep 'ms/step, ' gpurun_out/r06g/tb_f$f.txt)"
done
python -m pytest tests/test_gpu_train_step.py tests/test_gpu_train_b32.py tests/test_gpu_train_stages.py -x -q -m gpu > gpurun_out/r06g/t.log 2>&1; tail -3 gpurun_out/r06g/t.log
