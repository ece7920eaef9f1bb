#!/bin/bash
# Round 4: device timeline of score-msa on the configs[2] MAF with the device engine, as one group of blocks (kernel durations without
# another group's pruning launches beside them) and as the default two.  tools/r04_timeline.sh [out file]
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
W=/tmp/r04_work; mkdir -p $W/outp
MAF=$W/syn1mb.maf
[ -f $MAF ] || python3 tools/gen_synthetic_maf.py $MAF --bases 1000000 --seed 20261002 > /dev/null 2>&1 || exit 1
for s in omega mle; do
  for g in 1 2; do
    echo "== score-msa --strategy $s, PCSF_BLOCK_GROUPS=$g"
    rm -rf /tmp/tl_r04
    PCSF_FULL_TEARDOWN=1 PCSF_BLOCK_GROUPS=$g rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_r04 -- phylocsfpp_amd/host/phylocsf-amd score-msa --strategy $s --comp-anc 0 --output $W/outp tests/golden/models/58mammals $MAF 2>&1 | grep -E "blocks,"
    python3 tools/kernel_timeline.py /tmp/tl_r04 prune
  done
done
