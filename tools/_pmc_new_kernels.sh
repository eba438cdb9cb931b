# PMC passes (separate runs, as the pool requires) over the kernels added / rewritten in the second half of round 3: the weights-stationary HR conv
# (tools/bench_hr_conv.py) and the LayerNorm pair (tools/bench_ln.py).  Run on the GPU box from the repo root; writes gpurun_out/r03_j_pmc_*.txt
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
for tool in bench_hr_conv bench_ln; do
  for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    tag=$(echo $pmc | cut -d' ' -f1)
    rocprofv3 --pmc $pmc --output-format csv -d $O/pmc_${tool}_$tag -o p -- python3 tools/${tool}.py > /dev/null 2>&1
    if [ $tool = bench_hr_conv ]; then pat=conv_; else pat=layernorm; fi
    python tools/pmc_stats.py $(ls $O/pmc_${tool}_$tag/*counter_collection.csv | head -1) $pat > $O/r03_j_pmc_${tool}_$tag.txt
    rm -rf $O/pmc_${tool}_$tag
  done
done
cat $O/r03_j_pmc_bench_hr_conv_FETCH_SIZE.txt $O/r03_j_pmc_bench_hr_conv_WRITE_SIZE.txt
