set -e
python bench.py --workload train_vimeo --steps 4 > gpurun_out/r03_a_bench_vimeo_bf16.json 2> gpurun_out/r03_a_vimeo_bf16.err
python bench.py --workload train_vimeo --steps 4 --fp8 > gpurun_out/r03_a_bench_vimeo_fp8.json 2> gpurun_out/r03_a_vimeo_fp8.err
python bench.py --workload train_full --steps 8 > gpurun_out/r03_a_bench_full.json 2> gpurun_out/r03_a_full.err
python bench.py --workload train_full --steps 8 --graph --no-cpu-baseline > gpurun_out/r03_a_bench_full_graph.json 2> gpurun_out/r03_a_full_graph.err
python bench.py --workload train_swin --steps 6 > gpurun_out/r03_a_bench_swin.json 2> gpurun_out/r03_a_swin.err
python bench.py --workload infer --steps 2 --warmup 1 > gpurun_out/r03_a_bench_infer.json 2> gpurun_out/r03_a_infer.err
python bench.py --workload infer --steps 2 --warmup 1 --fp8 --no-cpu-baseline > gpurun_out/r03_a_bench_infer_fp8.json 2> gpurun_out/r03_a_infer_fp8.err
python bench.py --steps 8 --fp8 --no-cpu-baseline > gpurun_out/r03_a_bench_train_fp8.json 2> gpurun_out/r03_a_train_fp8.err
