#!/bin/bash
# A/B of PCSF_BLOCK_GROUPS on the configs[2] block leg: tools/ab_groups_c3.sh
cd "$(dirname "$0")/.."
for g in 2 3 4; do for rep in 1 2; do
  echo "== PCSF_BLOCK_GROUPS=$g (run $rep)"
  PCSF_BLOCK_GROUPS=$g bash tools/c3_quick.sh mle omega 2>&1 | grep "seconds: total"
done; done
