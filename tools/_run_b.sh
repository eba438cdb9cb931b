set -e
python -m pytest tests/test_grad_gpu.py tests/test_tab_kernels_gpu.py tests/test_modules_gpu.py tests/test_model_gpu.py -m gpu -q -x > gpurun_out/r3b_tests.log 2>&1
python bench.py --steps 8 --no-cpu-baseline > gpurun_out/r3b_bench_train.json 2> gpurun_out/r3b_bench_train.err
python bench.py --workload train_full --steps 8 --no-cpu-baseline > gpurun_out/r3b_bench_full.json 2> gpurun_out/r3b_bench_full.err
python bench.py --workload train_full --steps 8 --no-cpu-baseline --graph > gpurun_out/r3b_bench_full_graph.json 2> gpurun_out/r3b_bench_full_graph.err
