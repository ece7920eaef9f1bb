#!/bin/bash
# Round 5, after the eigen-solver changed: does the number of concurrent block groups (PCSF_BLOCK_GROUPS; default 1 below 1,536 blocks of OMEGA / 512 of
# MLE, else 2) still sit at its best?  The reference's 516 real blocks x 1 / 4 / 16 and BASELINE.json configs[2].
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
W=/tmp/r05_work
mkdir -p $W/out
MAF=$W/syn1mb.maf
make -s -C phylocsfpp_amd/host || exit 1
[ -f $MAF ] || python3 tools/gen_synthetic_maf.py $MAF --bases 1000000 --seed 20261002 > /dev/null 2>&1 || exit 1
CLI=phylocsfpp_amd/host/phylocsf-amd
for r in 1 4 16; do
  for g in default 1 2 3; do
    if [ $g = default ]; then unset PCSF_BLOCK_GROUPS; else export PCSF_BLOCK_GROUPS=$g; fi
    echo "== real blocks x $r, groups $g: $(python3 tools/bench_blocks.py - 0 $r nocpu 2>&1 | grep -E "GPU (MLE|OMEGA)" | tr -s ' ' | cut -d' ' -f2,3,4 | tr '\n' ' ')"
  done
done
for s in mle omega; do
  for g in default 1 2 3 4; do
    if [ $g = default ]; then unset PCSF_BLOCK_GROUPS; else export PCSF_BLOCK_GROUPS=$g; fi
    echo "== configs[2] $s, groups $g: $(GPU_MAX_HW_QUEUES=8 PCSF_TIME_BLOCKS=1 $CLI score-msa --strategy $s --comp-anc 0 --output $W/out tests/golden/models/58mammals $MAF 2>&1 | grep "scoring backend" | sed 's/.*scoring backend \([0-9.]*\).*/\1 s/')"
  done
done
