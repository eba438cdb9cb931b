# Fourth profiling pass of round 4: the inference workload's kernels (eager), for the graph-mode roofline constant.   bash tools/_prof_r04d.sh   (GPU box, repo root)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04_prof -o t -- python3 bench.py --workload infer --steps 1 --warmup 0 --no-cpu-baseline --no-prof > /dev/null 2>&1
T=$(ls $O/r04_prof/*kernel_trace.csv | head -1)
python3 tools/kernel_shapes.py $T conv_ > $O/r04_e_infer_conv_launches.txt
head -25 $(ls $O/r04_prof/*kernel_stats.csv | head -1) | cut -c1-220 > $O/r04_e_infer_kernel_stats_top.csv
rm -rf $O/r04_prof
head -6 $O/r04_e_infer_conv_launches.txt
