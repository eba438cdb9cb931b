#!/bin/bash
# A/B of two checkouts on ONE box (boxes of the pool differ by +-2 %): tools/ab_bench.sh <dirA> <dirB> [bench.py arguments]
# runs bench.py of A, B, A, B and prints value / ms_per_step of each run.  The usual pair: `git worktree add .ab_prev HEAD` (git-ignored, built with
# `python -m vmg_amd.build` inside it, travels to the GPU box with the snapshot) against the working tree: tools/ab_bench.sh .ab_prev .
A=$1; B=$2; shift 2
for d in $A $B $A $B; do
  (cd $d && python bench.py --steps 8 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('$d', d['value'], d['ms_per_step'])")
done
