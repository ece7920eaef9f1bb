#!/bin/bash
# Runs on the GPU box: old against new prune_factored (tools/_build/lib_factold.so = the previous commit's factored.hip in this tree's
# library), alternating, on the reference's real blocks.  bash tools/fact_ab2.sh <out dir>
cd ${GRAFT_REPO_ROOT:-.}; O=${1:-gpurun_out/fact_ab2}; mkdir -p $O
for r in 4 16; do for rep in 1 2; do for v in new old; do
  L=""; [ $v = old ] && L=$PWD/tools/_build/lib_factold.so
  echo "== $v, replicas $r, run $rep"; PCSF_LIBRARY=$L python3 tools/bench_blocks.py - 0 $r nocpu 2>&1 | grep -E "GPU (MLE|OMEGA)"
done; done; done > $O/ab.txt 2>&1
cat $O/ab.txt
