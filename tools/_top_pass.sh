#!/bin/bash
# three rocprofv3 passes over two bench steps (kernel trace, FETCH_SIZE, WRITE_SIZE) for tools/top_kernels.py; run on the GPU box from the repo root
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/top
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof > $O/trace.log 2>&1
echo trace done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof > $O/fetch.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof > $O/write.log 2>&1
echo write done
T=$(ls $O/trace/*/*kernel_trace.csv); F=$(ls $O/fetch/*/*counter_collection.csv); W=$(ls $O/write/*/*counter_collection.csv)
python tools/top_kernels.py $T $F $W $F 45 > $O/top.txt
rm -rf $O/trace $O/fetch $O/write
cat $O/top.txt
