#!/bin/bash
# Runs on the GPU box: the measurements DESIGN.md quotes for round 5.  bash tools/evidence_r05.sh [tag]  -> gpurun_out/<tag>/
# Every step's exit code goes to <out>/steps.log (a failed step must not leave a stale artefact unnoticed).
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-evidence_r05}; mkdir -p "$O"
: > "$O/steps.log"
step() { local name=$1; shift; "$@"; local rc=$?; echo "$name rc=$rc" >> "$O/steps.log"; return $rc; }
step make-host make -s -C phylocsfpp_amd/host
# ---- the bench line, its kernel statistics and PMC passes (separate runs)
step bench bash -c "python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err"
step bench-gather-rehearsal bash -c "python3 bench.py --force-gather 1 --steps 5 --end-to-end-runs 0 --block-runs 0 --cpu-baseline-sites 0 --other-models 0 > $O/bench_force_gather.json 2> $O/bench_force_gather.err"
export PCSF_FULL_TEARDOWN=1
step profile-bench bash -c "bash tools/profile_gpu.sh r05 > $O/profile_gpu.log 2>&1"
step copy-pmc cp gpurun_out/prof_r05/pmc_summary.json "$O/pmc_summary.json"
step copy-stats bash -c "cp \$(find gpurun_out/prof_r05/trace -name '*kernel_stats.csv' | head -1) $O/kernel_stats_bench.csv"
unset PCSF_FULL_TEARDOWN
# ---- round 5's A/B files
step two-tile bash -c "tools/_build/two_tile_bench > $O/two_tile_microbench.txt 2>&1"
step maf-direct bash -c "bash tools/maf_direct_ab.sh > $O/maf_direct_ab.txt 2>&1"
step factored-compiled bash -c "bash tools/factored_compiled_ab.sh > $O/factored_compiled_ab.txt 2>&1"
# ---- the eigen-solver's forms (second half of round 5: the seven-wave form became the default) and advance_blocks
mkdir -p $O/eigen_forms
step eigen-pmc bash -c "FORMS='quad reg' bash tools/eigen_pmc.sh > $O/eigen_forms/pmc_seven_wave_vs_two_wave.txt 2>&1"
step eigen-ab bash -c "{ FORMS='quad reg lds' bash tools/eigen_ab.sh; EIGEN_JOBS=512 FORMS='quad reg' bash tools/eigen_ab.sh; EIGEN_JOBS=2048 FORMS='quad reg' bash tools/eigen_ab.sh; } > $O/eigen_forms/ab_final.txt 2>&1"
step eigen-omega-ab bash -c "bash tools/eigen_form_omega_ab.sh > $O/eigen_forms/omega_end_to_end_ab.txt 2>&1"
step eigen-config3 bash -c "bash tools/r05_omega_forms.sh $O/eigen_forms > $O/eigen_forms/config3.log 2>&1"
step eigen-small-scale bash -c "python3 tools/eigen_form_small_scale.py 2>&1 | grep -v amdgpu.ids > $O/eigen_forms/objective_small_scales.txt"
step real-blocks bash -c "for r in 1 4 16; do echo \"== bench_blocks.py - 0 \$r nocpu\"; PCSF_TIME_BLOCKS=1 python3 tools/bench_blocks.py - 0 \$r nocpu 2>&1 | grep -E 'GPU |pruning kernel|run_omega: [0-9]+ blocks|run_mle: [0-9]+ blocks'; done > $O/real_blocks_final.txt 2>&1"
step omega-timeline bash -c "bash tools/omega_round_timeline.sh > $O/omega_round_timeline_516.txt 2>&1"
[ -f tools/_build/lib_advstamps.so ] && step advance-stamps bash -c "PCSF_LIBRARY=\$PWD/tools/_build/lib_advstamps.so python3 tools/bench_blocks.py - 0 1 nocpu 2>&1 | grep advance_blocks | awk 'NR%10==0' > $O/advance_blocks_stamps.txt"
[ -f tools/_build/lib_eigphase.so ] && step eigen-phases bash -c "PCSF_LIBRARY=\$PWD/tools/_build/lib_eigphase.so python3 tools/eigen_bench.py 200 2>&1 | grep 'eigen phase' | sed -n '13,40p' > $O/eigen_forms/prologue_epilogue_phases.txt"
# ---- BASELINE.json configs[3] on one GPU (100 Mb, text and packed): stage 1 with the thread pool and the vector scanners
step config4 bash -c "bash tools/config4_one_gpu.sh tmp_c4 100 > /dev/null 2>&1; mkdir -p $O/config4; cp gpurun_out/tmp_c4/config4/* $O/config4/; rm -rf gpurun_out/tmp_c4"
cat "$O/steps.log"
ls -la "$O"
