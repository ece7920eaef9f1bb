#!/bin/bash
# Runs on the GPU box: the measurements DESIGN.md quotes for round 4.  bash tools/evidence_r04.sh [tag]  -> gpurun_out/<tag>/
# Every step's exit code goes to <out>/steps.log (a failed step must not leave a stale artefact unnoticed).
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-evidence_r04}; mkdir -p "$O"
: > "$O/steps.log"
step() { local name=$1; shift; "$@"; local rc=$?; echo "$name rc=$rc" >> "$O/steps.log"; return $rc; }
step make-host make -s -C phylocsfpp_amd/host
# ---- the bench line (with the block legs' and the other models' rooflines), its kernel statistics and PMC passes (separate runs)
step bench bash -c "python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err"
export PCSF_FULL_TEARDOWN=1   # the CLI exits without tearing the runtime down; rocprofv3 writes its files at exit
step profile-bench bash -c "bash tools/profile_gpu.sh r04 > $O/profile_gpu.log 2>&1"
step copy-pmc cp gpurun_out/prof_r04/pmc_summary.json "$O/pmc_summary.json"
step copy-stats bash -c "cp \$(find gpurun_out/prof_r04/trace -name '*kernel_stats.csv' | head -1) $O/kernel_stats_bench.csv"
# ---- BASELINE.json configs[2], block strategies: engines side by side, groups, real blocks (tools/r04_blocks.sh), kernel statistics, PMC, timeline
step blocks bash -c "tools/r04_blocks.sh $O/blocks > $O/blocks.log 2>&1"
for s in mle omega; do
  step pmc-$s bash -c "bash tools/c3_pmc.sh $s $O/config3_${s}_pmc.json > $O/config3_${s}_pmc.txt 2>&1"
  rm -rf /tmp/prof_c3_$s; mkdir -p /tmp/r04_work/outp
  step stats-$s bash -c "rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3_$s -- phylocsfpp_amd/host/phylocsf-amd score-msa --strategy $s --comp-anc 0 --output /tmp/r04_work/outp tests/golden/models/58mammals /tmp/r04_work/syn1mb.maf > /dev/null 2>&1 && cp \$(find /tmp/prof_c3_$s -name '*kernel_stats.csv' | head -1) $O/config3_${s}_kernel_stats.csv"
done
step timeline bash -c "tools/r04_timeline.sh > $O/config3_timeline.txt 2>&1"
step round-trace bash -c "PCSF_BLOCK_GROUPS=1 PCSF_TRACE_ROUNDS=1 phylocsfpp_amd/host/phylocsf-amd score-msa --strategy omega --comp-anc 0 --output /tmp/r04_work/outp tests/golden/models/58mammals /tmp/r04_work/syn1mb.maf 2>&1 | grep '^round\|blocks,' > $O/config3_omega_round_trace.txt"
step blocks516-timeline bash -c "bash tools/blocks516_timeline.sh > $O/blocks516_timeline.txt 2>&1"
step tile-kernel bash -c "{ for a in '16384 64' '65536 64' '16384 64 100vertebrates'; do bash tools/tile_prof.sh \$a; done; } > $O/tile_kernel_sustained.txt 2>&1"
# ---- process start-up (the CLI's fast exit: no tear-down of the runtime)
unset PCSF_FULL_TEARDOWN
step startup bash -c "bash tools/startup_profile.sh > $O/startup_profile.txt 2>&1"
step startup-teardown bash -c "PCSF_FULL_TEARDOWN=1 bash tools/startup_profile.sh 2>&1 | grep -A6 'five runs' > $O/startup_profile_full_teardown.txt"
# ---- BASELINE.json configs[3] on one GPU (100 Mb, text and packed) and configs[4] at 20 Mb
step config4 bash -c "bash tools/config4_one_gpu.sh tmp_c4 100 > /dev/null 2>&1; mkdir -p $O/config4; cp gpurun_out/tmp_c4/config4/* $O/config4/; rm -rf gpurun_out/tmp_c4"
step wig_profile bash -c "bash tools/r04_wig_profile.sh tmp_wig > /dev/null 2>&1; mkdir -p $O/wig_profile; cp gpurun_out/tmp_wig/wig_profile/{kernel_stats.csv,run_wig0.txt,run_wig1.txt} $O/wig_profile/; rm -rf gpurun_out/tmp_wig"
step config5 bash -c "bash tools/config5_one_gpu.sh $O/c5 20 > /dev/null 2>&1; cp $O/c5/config5.txt $O/config5_one_gpu.txt; rm -rf $O/c5"
cat "$O/steps.log"
ls -la "$O"
