#!/bin/bash
# A/B of pcsf_prepare_block_scoring on the configs[2] block leg (same box, alternating): tools/ab_prepare.sh
cd "$(dirname "$0")/.."
for rep in 1 2 3; do for p in 1 0; do
  echo "== PCSF_PREPARE_BLOCKS=$p (run $rep)"
  PCSF_PREPARE_BLOCKS=$p bash tools/c3_quick.sh mle omega 2>&1 | grep "seconds: total"
done; done
