#!/bin/bash
# A/B of PCSF_BLOCK_GROUPS (and the HW-queue count) on the reference's 516-block file and 4 copies of it: tools/ab_groups516.sh
cd "$(dirname "$0")/.."
for copies in 1 4; do for q in 4 8; do for g in 1 2; do
  echo "== copies=$copies GPU_MAX_HW_QUEUES=$q PCSF_BLOCK_GROUPS=$g"
  for rep in 1 2; do GPU_MAX_HW_QUEUES=$q PCSF_BLOCK_GROUPS=$g python3 tools/bench_blocks.py - 0 $copies nocpu 2>&1 | grep "GPU OMEGA"; done
done; done; done
