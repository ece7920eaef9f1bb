#!/bin/bash
# device timeline of score-msa on the configs[2] MAF: tools/c3_timeline.sh <strategy> (env passed through)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
W=/tmp/config3_work; mkdir -p $W/outp
MAF=$W/syn1mb.maf
[ -f $MAF ] || python3 tools/gen_synthetic_maf.py $MAF --bases 1000000 --seed 20261002 > /dev/null 2>&1
rm -rf /tmp/tl_c3
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_c3 -- phylocsfpp_amd/host/phylocsf-amd score-msa --strategy $1 --output $W/outp tests/golden/models/58mammals $MAF 2>&1 | grep -E "blocks,"
python3 tools/kernel_timeline.py /tmp/tl_c3 ${2:-prune}
