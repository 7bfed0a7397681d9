mkdir -p gpurun_out/r06k
for i in 1 2; do
  python tools/train_bench.py --batch 32 --graph --steps 10 > gpurun_out/r06k/tb_$i.txt 2>&1
  echo "run=$i rc=$? $(grep 'ms/step, ' gpurun_out/r06k/tb_$i.txt)"
done
python -m pytest tests/test_gpu_train_step.py tests/test_gpu_train_b32.py tests/test_gpu_train_stages.py -x -q -m gpu > gpurun_out/r06k/t.log 2>&1; tail -3 gpurun_out/r06k/t.log
