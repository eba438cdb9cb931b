# rocprofv3 records behind profiles/r04_* (round 4).  Run on the GPU box from the repo root: bash tools/_prof_r04.sh
#  1. kernel trace of the default bench: kernel stats, the last step's kernel list, the per-grid summary, the conv launches by kernel and grid;
#  2. kernel trace of the full-config shard (eager): its step by kernel (are the torch pad / permute copies of the MorphFC branches gone?);
#  3. PMC passes (each its own run, as the pool requires) over the two 3x3 weight-gradient kernels (tools/bench_wgrad3_ab.py) and the dominant
#     convolution alone (tools/k1_traffic.py: FETCH_SIZE, WRITE_SIZE, MFMA busy);
#  4. the default bench line itself.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04_prof -o t -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras > $O/r04_bench_under_prof.json 2>/dev/null
T=$(ls $O/r04_prof/*kernel_trace.csv | head -1)
cp $(ls $O/r04_prof/*kernel_stats.csv | head -1) $O/r04_bench_kernel_stats.csv
python3 tools/step_kernels.py $T 60 > $O/r04_step_kernels.txt
python3 tools/prof_summary.py $T 66 $O/r04_bench_laststep_summary.txt
python3 tools/kernel_shapes.py $T conv_ > $O/r04_conv_launches.txt
rm -rf $O/r04_prof
echo "[prof] default bench traced"
rocprofv3 --kernel-trace --output-format csv -d $O/r04_prof -o t -- python3 bench.py --workload train_full --steps 3 --warmup 2 --no-cpu-baseline --no-prof > $O/r04_bench_full_under_prof.json 2>/dev/null
T=$(ls $O/r04_prof/*kernel_trace.csv | head -1)
python3 tools/step_kernels.py $T 70 > $O/r04_full_step_kernels.txt
rm -rf $O/r04_prof
echo "[prof] full-config shard traced"
for pmc in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $pmc | cut -d' ' -f1)
  rocprofv3 --pmc $pmc --output-format csv -d $O/pmc_w3_$tag -o p -- python3 tools/bench_wgrad3_ab.py 2 > /dev/null 2>&1
  python3 tools/pmc_stats.py $(ls $O/pmc_w3_$tag/*counter_collection.csv | head -1) conv_wgrad3 > $O/r04_pmc_wgrad3_$tag.txt
  rm -rf $O/pmc_w3_$tag
done
echo "[prof] wgrad3 counters done"
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $pmc | cut -d' ' -f1)
  rocprofv3 --pmc $pmc --output-format csv -d $O/pmc_k1_$tag -o p -- python3 tools/k1_traffic.py > /dev/null 2>&1
  python3 tools/pmc_stats.py $(ls $O/pmc_k1_$tag/*counter_collection.csv | head -1) conv_ws > $O/r04_pmc_k1_$tag.txt
  rm -rf $O/pmc_k1_$tag
done
echo "[prof] dominant kernel counters done"
python3 bench.py --steps 20 --warmup 5 > $O/r04_bench_train.json 2> $O/r04_bench_train.err
head -c 400 $O/r04_bench_train.json
