# Third profiling pass of round 4: the full-config shard (cfg3, one clip per GPU) on the current code.   bash tools/_prof_r04c.sh   (GPU box, repo root)
#  kernel trace of --workload train_full (eager) -> r04_e_full_step_kernels.txt, r04_e_full_launches.txt (every kernel by grid), and the graph line
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
rocprofv3 --kernel-trace --output-format csv -d $O/r04_prof -o t -- python3 bench.py --workload train_full --steps 3 --warmup 2 --no-cpu-baseline --no-prof > /dev/null 2>&1
T=$(ls $O/r04_prof/*kernel_trace.csv | head -1)
python3 tools/step_kernels.py $T 80 > $O/r04_e_full_step_kernels.txt
python3 tools/kernel_shapes.py $T conv_ > $O/r04_e_full_conv_launches.txt
python3 tools/prof_summary.py $T 80 $O/r04_e_full_laststep_summary.txt
rm -rf $O/r04_prof
python3 bench.py --workload train_full --steps 8 --warmup 3 --graph > $O/r04_e_bench_full_graph.json 2>/dev/null
head -c 300 $O/r04_e_bench_full_graph.json
