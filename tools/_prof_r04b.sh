# Second profiling pass of round 4 (final code state): bash tools/_prof_r04b.sh   (GPU box, repo root)
#  1. kernel trace of the default bench -> r04_c_step_kernels.txt, r04_c_bench_laststep_summary.txt, r04_c_conv_launches.txt, kernel stats
#  2. kernel trace of --workload train_swin -> r04_c_swin_step_kernels.txt (the MFMA window-attention kernels inside a step)
#  3. PMC passes over tools/bench_win3d.py (MFMA busy, VALU instructions) -> r04_c_pmc_win3d_*.txt
#  4. the bench lines: default (driver's command), train_swin, train_vimeo (bf16 and --fp8), infer (--graph)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04_prof -o t -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
T=$(ls $O/r04_prof/*kernel_trace.csv | head -1)
cp $(ls $O/r04_prof/*kernel_stats.csv | head -1) $O/r04_c_bench_kernel_stats.csv
python3 tools/step_kernels.py $T 60 > $O/r04_c_step_kernels.txt
python3 tools/prof_summary.py $T 66 $O/r04_c_bench_laststep_summary.txt
python3 tools/kernel_shapes.py $T conv_ > $O/r04_c_conv_launches.txt
rm -rf $O/r04_prof
echo "[prof] default bench traced"
rocprofv3 --kernel-trace --output-format csv -d $O/r04_prof -o t -- python3 bench.py --workload train_swin --steps 3 --warmup 2 --no-cpu-baseline --no-prof > /dev/null 2>&1
T=$(ls $O/r04_prof/*kernel_trace.csv | head -1)
python3 tools/step_kernels.py $T 60 > $O/r04_c_swin_step_kernels.txt
rm -rf $O/r04_prof
echo "[prof] train_swin traced"
for pmc in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS"; do
  tag=$(echo $pmc | cut -d' ' -f1)
  rocprofv3 --pmc $pmc --output-format csv -d $O/pmc_w3d_$tag -o p -- python3 tools/bench_win3d.py 3 > /dev/null 2>&1
  python3 tools/pmc_stats.py $(ls $O/pmc_w3d_$tag/*counter_collection.csv | head -1) win3d > $O/r04_c_pmc_win3d_$tag.txt
  rm -rf $O/pmc_w3d_$tag
done
echo "[prof] win3d counters done"
python3 bench.py --steps 20 --warmup 5 > $O/r04_c_bench_train.json 2> $O/r04_c_bench_train.err
echo "[prof] default line done"
python3 bench.py --workload train_swin --steps 8 --warmup 3 > $O/r04_c_bench_swin.json 2>/dev/null
python3 bench.py --workload train_vimeo --steps 3 --warmup 2 > $O/r04_c_bench_vimeo_bf16.json 2>/dev/null
echo "[prof] vimeo bf16 done"
python3 bench.py --workload train_vimeo --steps 3 --warmup 2 --fp8 > $O/r04_c_bench_vimeo_fp8.json 2>/dev/null
echo "[prof] vimeo fp8 done"
python3 bench.py --workload infer --steps 1 --warmup 1 --graph > $O/r04_c_bench_infer_graph.json 2>/dev/null
python3 bench.py --workload train_full --steps 8 --warmup 3 --graph > $O/r04_c_bench_full_graph.json 2>/dev/null
for f in train swin vimeo_bf16 vimeo_fp8 infer_graph full_graph; do head -c 260 $O/r04_c_bench_$f.json | tail -c 140; echo; done
