# rocprofv3 records behind profiles/r03_*: kernel trace of the default bench, per-grid summary of the last step, PMC passes on the two dominant conv kernels
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_prof -o t -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r03_f_bench_under_prof.json 2>/dev/null
cp $(ls gpurun_out/r03_prof/*kernel_stats.csv | head -1) gpurun_out/r03_f_bench_kernel_stats.csv
python tools/step_kernels.py $(ls gpurun_out/r03_prof/*kernel_trace.csv | head -1) 60 > gpurun_out/r03_f_step_kernels.txt
python tools/prof_summary.py $(ls gpurun_out/r03_prof/*kernel_trace.csv | head -1) > gpurun_out/r03_f_bench_laststep_summary.txt 2>/dev/null || true
rm -rf gpurun_out/r03_prof
python bench.py --steps 8 > gpurun_out/r03_f_bench_train.json 2> gpurun_out/r03_f_bench_train.err
for tool in k1 q8; do
  for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "GRBM_GUI_ACTIVE"; do
    tag=$(echo $pmc | cut -d' ' -f1)
    rocprofv3 --pmc $pmc --output-format csv -d gpurun_out/r03_pmc_${tool}_$tag -o p -- python3 tools/${tool}_traffic.py > /dev/null 2>&1
    python tools/pmc_stats.py $(ls gpurun_out/r03_pmc_${tool}_$tag/*counter_collection.csv | head -1) conv > gpurun_out/r03_g_${tool}_pmc_$tag.txt
    rm -rf gpurun_out/r03_pmc_${tool}_$tag
  done
done
