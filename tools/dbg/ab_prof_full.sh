cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4w
mkdir -p $O
for d in .ab_prev .; do
  cd $GRAFT_REPO_ROOT/$d
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o t -- python3 bench.py --workload train_full --steps 3 --warmup 2 --no-cpu-baseline --no-prof > /dev/null 2>&1
  S=$(ls $O/p/*kernel_stats.csv | head -1)
  echo "== $d" >> $O/stats.txt
  grep -E "pack_batch|conv_ksplit|conv_pack" $S | cut -c1-200 >> $O/stats.txt
  rm -rf $O/p
done
cat $O/stats.txt
