#!/bin/bash
# BASELINE.json configs[2], the block-strategy half: `score-msa --strategy mle|omega` with the 58mammals model on the synthetic
# 100-species 1 Mb MAF (tools/gen_synthetic_maf.py --bases 1000000 --seed 20261002; reference: src/run.hpp:59-182 called per block at
# src/phylocsf++score_msa.hpp:94-127).  Runs on the GPU box:  tools/config3_blocks.sh [out dir] [prof]
#   timings of the CLI (PCSF_TIME_BLOCKS=1: rounds, evaluations, path split), and with `prof` the rocprofv3 kernel statistics of both
#   strategies and the oracle driver on a prefix of the same file.
cd "$(dirname "$0")/.."
ROOT=$PWD
O=${1:-gpurun_out/config3}
mkdir -p $O
export TMPDIR=/tmp
W=/tmp/config3_work
mkdir -p $W
MAF=$W/syn1mb.maf
[ -f $MAF ] || python3 tools/gen_synthetic_maf.py $MAF --bases 1000000 --seed 20261002 > $O/maf.txt 2>&1
CLI=phylocsfpp_amd/host/phylocsf-amd
MODEL=${PCSF_MODEL_PREFIX:-tests/golden/models/58mammals}
{
for s in fixed mle omega; do
  for rep in 1 2; do
    echo "== score-msa --strategy $s (run $rep)"
    mkdir -p $W/out_$s
    TIMEFORMAT="process wall %R s"
    time PCSF_TIME_BLOCKS=1 $CLI score-msa --strategy $s --comp-anc $([ $s = omega ] && echo 0 || echo 1) --output $W/out_$s $MODEL $MAF 2>&1 | grep -v "^start-up"
  done
done
} > $O/score_msa_config3.txt 2>&1
cat $O/score_msa_config3.txt
if [ "$2" = "prof" ]; then
  for s in mle omega; do
    mkdir -p $W/outp_$s
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3_$s -- $CLI score-msa --strategy $s --output $W/outp_$s $MODEL $MAF > /dev/null 2>&1
    cp $(find /tmp/prof_c3_$s -name "*kernel_stats.csv" | head -1) $O/config3_${s}_kernel_stats.csv
    echo "== kernel statistics, $s"; head -8 $O/config3_${s}_kernel_stats.csv | cut -c1-170
  done
fi
