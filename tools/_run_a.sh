set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 8 --no-cpu-baseline > gpurun_out/r3a_bench_train.json 2> gpurun_out/r3a_bench_train.err
python bench.py --workload train_full --steps 8 --no-cpu-baseline > gpurun_out/r3a_bench_full.json 2> gpurun_out/r3a_bench_full.err
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3a_prof_full -o full -- python bench.py --workload train_full --steps 3 --warmup 2 --no-cpu-baseline --no-prof > gpurun_out/r3a_under_prof_full.json 2>/dev/null
python tools/step_kernels.py $(ls gpurun_out/r3a_prof_full/*kernel_trace.csv | head -1) 60 > gpurun_out/r3a_full_step_kernels.txt
rm -rf gpurun_out/r3a_prof_full
