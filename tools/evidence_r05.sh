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
step eigen-pmc bash -c "FORMS=reg bash tools/eigen_pmc.sh > $O/eigen_pmc_product_form.txt 2>&1"
# ---- BASELINE.json configs[3] on one GPU (100 Mb, text and packed): stage 1 with the thread pool and the vector scanners
step config4 bash -c "bash tools/config4_one_gpu.sh tmp_c4 100 > /dev/null 2>&1; mkdir -p $O/config4; cp gpurun_out/tmp_c4/config4/* $O/config4/; rm -rf gpurun_out/tmp_c4"
cat "$O/steps.log"
ls -la "$O"
