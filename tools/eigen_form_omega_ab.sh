#!/bin/bash
# OMEGA end to end under the two register-resident forms of the eigen-solver (PCSF_EIGEN_FORM=1: two waves, 2: seven waves), alternating on
# one box: tools/bench_blocks.py on the reference's 516 real blocks x 1 / 4 / 16.
cd "$(dirname "$0")/.." || exit 1
for r in ${REPLICAS:-1 4 16}; do
  for rep in 1 2; do
    for f in 1 2; do
      echo "== form $f, replicas $r, run $rep: $(PCSF_EIGEN_FORM=$f python3 tools/bench_blocks.py - 0 $r nocpu 2>&1 | grep -E "GPU OMEGA" | tr -s ' ')"
    done
  done
done
