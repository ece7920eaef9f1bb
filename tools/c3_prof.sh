#!/bin/bash
# rocprofv3 kernel statistics of score-msa on the configs[2] MAF: tools/c3_prof.sh <strategy> <out csv> (env passed through)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
W=/tmp/config3_work; mkdir -p $W/outp
MAF=$W/syn1mb.maf
[ -f $MAF ] || python3 tools/gen_synthetic_maf.py $MAF --bases 1000000 --seed 20261002 > /dev/null 2>&1
rm -rf /tmp/prof_c3
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -- phylocsfpp_amd/host/phylocsf-amd score-msa --strategy $1 --output $W/outp tests/golden/models/58mammals $MAF > /dev/null 2>&1
mkdir -p $(dirname $2)
cp $(find /tmp/prof_c3 -name "*kernel_stats.csv" | head -1) $2
head -8 $2 | cut -c1-150
