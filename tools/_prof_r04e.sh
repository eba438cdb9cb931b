# Fifth profiling pass of round 4 (final code state of the default bench): bash tools/_prof_r04e.sh   (GPU box, repo root)
#  kernel trace of the default bench -> r04_e_step_kernels.txt, r04_e_bench_laststep_summary.txt, r04_e_conv_launches.txt, kernel stats; then the driver's bench line
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04_prof -o t -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
T=$(ls $O/r04_prof/*kernel_trace.csv | head -1)
cp $(ls $O/r04_prof/*kernel_stats.csv | head -1) $O/r04_e_bench_kernel_stats.csv
python3 tools/step_kernels.py $T 70 > $O/r04_e_step_kernels.txt
python3 tools/prof_summary.py $T 66 $O/r04_e_bench_laststep_summary.txt
python3 tools/kernel_shapes.py $T conv_ > $O/r04_e_conv_launches.txt
rm -rf $O/r04_prof
python3 bench.py --steps 20 --warmup 5 > $O/r04_e_bench_train.json 2> $O/r04_e_bench_train.err
head -c 300 $O/r04_e_bench_train.json; echo; head -3 $O/r04_e_step_kernels.txt
