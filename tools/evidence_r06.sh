#!/bin/bash
# Runs on the GPU box: the measurements DESIGN.md quotes for round 6.  bash tools/evidence_r06.sh [tag]  -> gpurun_out/<tag>/
# Every step's exit code goes to <out>/steps.log (a failed step must not leave a stale artefact unnoticed).
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-evidence_r06}; mkdir -p "$O"
: > "$O/steps.log"
step() { local name=$1; shift; "$@"; local rc=$?; echo "$name rc=$rc" >> "$O/steps.log"; return $rc; }
step make-host make -s -C phylocsfpp_amd/host
# ---- the bench line (subtree tables are the default), its kernel statistics and PMC passes (separate runs)
step bench bash -c "python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err"
step bench-full-program bash -c "PCSF_CHERRY_TABLES=0 python3 bench.py --end-to-end-runs 0 --block-runs 0 --cpu-baseline-sites 0 > $O/bench_n1_full_program.json 2> /dev/null"
step bench-gather-rehearsal bash -c "python3 bench.py --force-gather 1 --steps 5 --end-to-end-runs 0 --block-runs 0 --cpu-baseline-sites 0 --other-models 0 > $O/bench_force_gather.json 2> $O/bench_force_gather.err"
export PCSF_FULL_TEARDOWN=1
step profile-bench bash -c "bash tools/profile_gpu.sh r06 > $O/profile_gpu.log 2>&1"
step copy-pmc cp gpurun_out/prof_r06/pmc_summary.json gpurun_out/prof_r06/traffic.json "$O/"
step copy-stats bash -c "cp \$(find gpurun_out/prof_r06/trace -name '*kernel_stats.csv' | head -1) $O/kernel_stats_bench.csv"
unset PCSF_FULL_TEARDOWN
# ---- the three forms of tracks mode side by side (full program, cherries as leaves, triples too), three models
step cherry-ab bash -c "bash tools/cherry_ab.sh > $O/subtree_tables_ab.txt 2>&1"
# ---- one process per GPU: the ranks command line rehearsed on this GPU (RCCL communicator of one rank)
step ranks-probe bash -c "bash tools/ranks_probe.sh 2>&1 | grep -v '^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|NCCL INFO Channel\|WARN Could not read node\|^\$' | head -70 > $O/ranks_probe.txt"
# ---- the block strategies on the reference's real blocks (unchanged kernels this round: the state of the box)
step real-blocks bash -c "for r in 1 16; do echo \"== bench_blocks.py - 0 \$r nocpu\"; PCSF_TIME_BLOCKS=1 python3 tools/bench_blocks.py - 0 \$r nocpu 2>&1 | grep -E 'GPU |pruning kernel|run_omega: [0-9]+ blocks|run_mle: [0-9]+ blocks'; done > $O/real_blocks.txt 2>&1"
# ---- BASELINE.json configs[3] on one GPU (100 Mb, text and packed)
step config4 bash -c "bash tools/config4_one_gpu.sh tmp_c4 100 > /dev/null 2>&1; mkdir -p $O/config4; cp gpurun_out/tmp_c4/config4/* $O/config4/; rm -rf gpurun_out/tmp_c4"
cat "$O/steps.log"
ls -la "$O"
