#!/bin/bash
# prune_factored with the tiles' op lists compiled once per call (round 5) against deriving them at every evaluation (PCSF_FACTORED_COMPILED=0):
# tools/bench_blocks.py on the reference's 516 real blocks x 1 / 4 / 16, alternating on one box; plus a tile's cycles (PCSF_FACTORED_STAMPS=1).
cd "$(dirname "$0")/.." || exit 1
for reps in 1 4 16; do
    for c in 1 0 1 0; do
        echo "== 516 real blocks x $reps, PCSF_FACTORED_COMPILED=$c"
        PCSF_FACTORED_COMPILED=$c python3 tools/bench_blocks.py - 0 $reps nocpu 2>&1 | grep "GPU MLE\|GPU OMEGA"
    done
done
for c in 1 0; do
    echo "== stamps, 516 x 4, PCSF_FACTORED_COMPILED=$c"
    PCSF_FACTORED_COMPILED=$c PCSF_FACTORED_STAMPS=1 PCSF_TIME_BLOCKS=1 python3 tools/bench_blocks.py - 0 4 nocpu 2>&1 | grep -i "stamp\|cycles\|tile" | tail -6
done
