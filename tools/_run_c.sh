set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3c_prof -o t -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-prof > gpurun_out/r3c_under_prof.json 2>/dev/null
python tools/step_kernels.py $(ls gpurun_out/r3c_prof/*kernel_trace.csv | head -1) 70 > gpurun_out/r3c_step_kernels.txt
rm -rf gpurun_out/r3c_prof
