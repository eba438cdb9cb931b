# rocprofv3 records behind profiles/r03_j_* (end of round 3): kernel trace of the default bench (kernel stats, the last step's kernel list, the per-grid
# summary of the last 50 ms, the conv launches by kernel and grid), then the default bench line itself.  Run on the GPU box from the repo root.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_prof -o t -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/r03_j_bench_under_prof.json 2>/dev/null
T=$(ls $O/r03_prof/*kernel_trace.csv | head -1)
cp $(ls $O/r03_prof/*kernel_stats.csv | head -1) $O/r03_j_bench_kernel_stats.csv
python tools/step_kernels.py $T 60 > $O/r03_j_step_kernels.txt
python tools/prof_summary.py $T 66 $O/r03_j_bench_laststep_summary.txt  # (the step period under the tracer is ~66 ms)
python tools/kernel_shapes.py $T conv_ > $O/r03_j_conv_launches.txt
rm -rf $O/r03_prof
python bench.py > $O/r03_j_bench_train.json 2> $O/r03_j_bench_train.err
cat $O/r03_j_bench_train.json
