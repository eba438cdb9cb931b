# Sixth profiling pass of round 4: the Vimeo-size train step (configs[4]'s shape, bf16) by kernel.   bash tools/_prof_r04f.sh   (GPU box, repo root)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
rocprofv3 --kernel-trace --output-format csv -d $O/r04_prof -o t -- python3 bench.py --workload train_vimeo --steps 2 --warmup 1 --no-cpu-baseline --no-prof > /dev/null 2>&1
T=$(ls $O/r04_prof/*kernel_trace.csv | head -1)
python3 tools/step_kernels.py $T 50 > $O/r04_e_vimeo_step_kernels.txt
python3 tools/kernel_shapes.py $T conv_ > $O/r04_e_vimeo_conv_launches.txt
python3 tools/kernel_shapes.py $T linear_ > $O/r04_e_vimeo_linear_launches.txt
rm -rf $O/r04_prof
head -24 $O/r04_e_vimeo_step_kernels.txt; grep -v wgrad $O/r04_e_vimeo_conv_launches.txt | head -24; head -8 $O/r04_e_vimeo_linear_launches.txt
