#!/bin/bash
# quick look at the configs[2] block leg: tools/c3_quick.sh [strategies...]   (env is passed through)
cd "$(dirname "$0")/.."
W=/tmp/config3_work; mkdir -p $W/out
MAF=$W/syn1mb.maf
[ -f $MAF ] || python3 tools/gen_synthetic_maf.py $MAF --bases 1000000 --seed 20261002 > /dev/null 2>&1
for s in ${@:-mle omega}; do
  echo "== $s"
  PCSF_TIME_BLOCKS=1 phylocsfpp_amd/host/phylocsf-amd score-msa --strategy $s --output $W/out tests/golden/models/58mammals $MAF 2>&1 | grep -v "^start-up"
  md5sum $W/out/syn1mb.maf.scores
done
