#!/bin/bash
# A/B of two builds of the library on ONE box (boxes of the pool differ by +-2 %): tools/ab_bench.sh <libA.so> <libB.so> [bench.py arguments]
# runs bench.py A, B, A, B and prints value / ms_per_step of each run.  (VMG_HIP_LIB selects the library, vmg_amd/hip.py)
A=$1; B=$2; shift 2
for lib in $A $B $A $B; do
  VMG_HIP_LIB=$lib python bench.py --steps 8 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('$lib', d['value'], d['ms_per_step'])"
done
