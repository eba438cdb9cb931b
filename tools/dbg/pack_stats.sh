# rocprofv3 kernel stats of the weight-pack kernels inside the full-configuration shard's step (working tree).   bash tools/dbg/pack_stats.sh   (GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/packst
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o t -- python3 bench.py --workload ${1:-train_full} --steps 3 --warmup 2 --no-cpu-baseline --no-prof > /dev/null 2>&1
S=$(ls $O/p/*kernel_stats.csv | head -1)
grep -E "pack" $S | cut -c1-170
rm -rf $O/p
