#!/bin/bash
# Round 4: the MLE / OMEGA engines side by side (device = chain.hip's advance_blocks, host = the lock-step rounds on host threads) on
# BASELINE.json configs[2]'s synthetic 1 Mb MAF (58mammals, dense blocks -> prune_tiles) and on the reference's 516 real blocks
# (100vertebrates, sparse -> prune_factored), x1 / x4 / x16.  Runs on the GPU box:  tools/r04_blocks.sh [out dir] [groups...]
cd "$(dirname "$0")/.." || exit 1
O=${1:-gpurun_out/r04_blocks}
mkdir -p "$O"
export TMPDIR=/tmp
W=/tmp/r04_work
mkdir -p $W
MAF=$W/syn1mb.maf
[ -f $MAF ] || python3 tools/gen_synthetic_maf.py $MAF --bases 1000000 --seed 20261002 > "$O/maf.txt" 2>&1 || exit 1
CLI=phylocsfpp_amd/host/phylocsf-amd
{
for s in mle omega; do
  for engine in device host; do
    for rep in 1 $([ $engine = device ] && echo 2); do
      echo "== configs[2] score-msa --strategy $s, $engine engine (run $rep)"
      mkdir -p $W/out_${s}_$engine
      PCSF_BLOCK_ENGINE=$engine PCSF_TIME_BLOCKS=1 $CLI score-msa --strategy $s --comp-anc $([ $s = omega ] && echo 0 || echo 1) --output $W/out_${s}_$engine tests/golden/models/58mammals $MAF 2>&1 | grep -v "^start-up"
    done
  done
  cmp $W/out_${s}_device/syn1mb.maf.scores $W/out_${s}_host/syn1mb.maf.scores && echo "== $s: device and host engines wrote identical .scores files"
done
mkdir -p $W/out_g
for cs in 0 1 0 1; do
  for s in mle omega; do
  echo "== configs[2] $s, device engine, PCSF_CONTROL_STREAM=$cs"
  PCSF_CONTROL_STREAM=$cs PCSF_TIME_BLOCKS=1 $CLI score-msa --strategy $s --comp-anc 0 --output $W/out_g tests/golden/models/58mammals $MAF 2>&1 | grep "scoring backend"
  done
done
for s in mle omega; do
for g in 1 2 3 4; do
  echo "== configs[2] $s, device engine, PCSF_BLOCK_GROUPS=$g"
  GPU_MAX_HW_QUEUES=8 PCSF_BLOCK_GROUPS=$g PCSF_TIME_BLOCKS=1 $CLI score-msa --strategy $s --comp-anc 0 --output $W/out_g tests/golden/models/58mammals $MAF 2>&1 | grep "scoring backend"
done
done
} > "$O/config3_engines.txt" 2>&1
cat "$O/config3_engines.txt"
{
for reps in 1 4 16; do
  for engine in device $([ $reps = 1 ] && echo host); do
    echo "== 516 real blocks x $reps, $engine engine"
    PCSF_BLOCK_ENGINE=$engine PCSF_TIME_BLOCKS=1 python3 tools/bench_blocks.py - 0 $reps nocpu 2>&1 | grep "^GPU\|run_mle: [0-9]\|run_omega: [0-9]\|blocks,"
  done
done
} > "$O/real_blocks_engines.txt" 2>&1
cat "$O/real_blocks_engines.txt"

# kernel statistics of the OMEGA run (device engine)
mkdir -p $W/outp
rm -rf /tmp/prof_r04
PCSF_FULL_TEARDOWN=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r04 -- $CLI score-msa --strategy omega --comp-anc 0 --output $W/outp tests/golden/models/58mammals $MAF > /dev/null 2>&1
cp $(find /tmp/prof_r04 -name "*kernel_stats.csv" | head -1) "$O/config3_omega_kernel_stats.csv" && head -12 "$O/config3_omega_kernel_stats.csv" | cut -c1-200
