set -e
python -m pytest tests/test_modules_gpu.py tests/test_model_gpu.py tests/test_grad_gpu.py -m gpu -q -x -k "mlp_cnn or (reds_full and not bf16_tol) " > gpurun_out/r3d_tests.log 2>&1 || true
tail -3 gpurun_out/r3d_tests.log
python bench.py --workload train_full --steps 8 --no-cpu-baseline > gpurun_out/r3d_bench_full.json 2> gpurun_out/r3d_bench_full.err
python bench.py --workload train_full --steps 8 --no-cpu-baseline --graph > gpurun_out/r3d_bench_full_graph.json 2> gpurun_out/r3d_bench_full_graph.err
