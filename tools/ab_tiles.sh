#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
W=/tmp/config3_work; mkdir -p $W/out_t $W/out_f
MAF=$W/syn1mb.maf
python3 tools/gen_synthetic_maf.py $MAF --bases 1000000 --seed 20261002 > /dev/null 2>&1
CLI=phylocsfpp_amd/host/phylocsf-amd
for s in mle omega; do
  echo "== $s tiles"; PCSF_TIME_BLOCKS=1 timeout 300 $CLI score-msa --strategy $s --comp-anc $([ $s = mle ] && echo 1 || echo 0) --output $W/out_t tests/golden/models/58mammals $MAF 2>&1 | grep -v "^start-up"
  cp $W/out_t/syn1mb.maf.scores $W/out_t/$s.scores
  echo "== $s factored"; PCSF_BLOCK_KERNEL=factored PCSF_TIME_BLOCKS=1 timeout 300 $CLI score-msa --strategy $s --comp-anc $([ $s = mle ] && echo 1 || echo 0) --output $W/out_f tests/golden/models/58mammals $MAF 2>&1 | grep -v "^start-up"
  cp $W/out_f/syn1mb.maf.scores $W/out_f/$s.scores
  python3 tools/c3_compare.py $W/out_t/$s.scores $W/out_f/$s.scores
done
timeout 900 python3 -m pytest tests/test_gpu_blocks.py tests/test_gpu_parity_blocks.py -x -q -m gpu 2>&1 | tail -15
