#!/bin/bash
# Runs on the GPU box: the measurements DESIGN.md quotes for round 3.  bash tools/evidence_r03.sh [tag]  -> gpurun_out/<tag>/
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; O=gpurun_out/${1:-evidence_r03}; mkdir -p $O
make -s -C phylocsfpp_amd/host
# ---- the bench line, its kernel statistics and PMC passes (separate runs)
python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err
bash tools/profile_gpu.sh r03 > $O/profile_gpu.log 2>&1
cp gpurun_out/prof_r03/pmc_summary.json $O/pmc_summary.json
cp $(find gpurun_out/prof_r03/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats_bench.csv
# ---- BASELINE.json configs[2], block strategies: score-msa mle / omega on the synthetic 1 Mb MAF
bash tools/config3_blocks.sh $O prof > /dev/null 2>&1
for s in mle omega; do bash tools/c3_pmc.sh $s $O/config3_${s}_pmc.json > $O/config3_${s}_pmc.txt 2>&1; done
{ for s in mle omega; do echo "== score-msa --strategy $s"; bash tools/c3_timeline.sh $s; done; } > $O/config3_timeline.txt 2>&1
{ echo "== PCSF_BLOCK_KERNEL=factored (round 2's pair: prune_factored + expm / block-mode prune_fixed)"; PCSF_BLOCK_KERNEL=factored bash tools/c3_quick.sh mle omega 2>&1 | grep -E "seconds:|pruning kernel|run_";
  echo "== PCSF_OMEGA_HELD_SYSTEMS=0"; PCSF_OMEGA_HELD_SYSTEMS=0 bash tools/c3_quick.sh omega 2>&1 | grep -E "seconds:|run_omega: [0-9]+ blocks";
  echo "== PCSF_EAGER_RESTARTS_BELOW=1000000 (restarts evaluated ahead of need, as in round 2)"; PCSF_EAGER_RESTARTS_BELOW=1000000 bash tools/c3_quick.sh omega 2>&1 | grep -E "seconds:|run_omega: [0-9]+ "; } > $O/config3_ablations.txt 2>&1
# ---- the tile kernel alone: sustained rate, stamps
{ for a in "16384 64" "65536 64" "16384 64 100vertebrates"; do bash tools/tile_prof.sh $a; done; } > $O/tile_kernel_sustained.txt 2>&1
# ---- the reference's real blocks (516-block file and copies of it): both pruning kernels
{ for r in 1 4 16; do echo "== bench_blocks.py - 0 $r nocpu (default: kernel chosen per list)"; PCSF_TIME_BLOCKS=1 python3 tools/bench_blocks.py - 0 $r nocpu 2>&1 | grep -E "GPU |pruning kernel|run_omega: [0-9]+ blocks|run_mle: [0-9]+ blocks";
    echo "== PCSF_BLOCK_KERNEL=tiles"; PCSF_BLOCK_KERNEL=tiles python3 tools/bench_blocks.py - 0 $r nocpu 2>&1 | grep "GPU "; done; } > $O/blocks_real_files.txt 2>&1
bash tools/blocks516_timeline.sh > $O/blocks516_timeline.txt 2>&1
# ---- the eigen-solver kernel alone: 200 jobs per launch (cold / warm starts), and with parts of a step switched off (timing only; the
#      variants are built here: tools/build_variant.sh eigN eigen.hip -DPCSF_EIG_FIXED_ROUNDS=120 -DPCSF_EIG_ABL=N, eigR0: -DPCSF_EIG_FIXED_ROUNDS=0)
for v in 0 1 2 4 8 16 31; do bash tools/build_variant.sh eig$v eigen.hip -DPCSF_EIG_FIXED_ROUNDS=120 -DPCSF_EIG_ABL=$v > /dev/null 2>&1 & done
bash tools/build_variant.sh eigR0 eigen.hip -DPCSF_EIG_FIXED_ROUNDS=0 > /dev/null 2>&1; wait
{ echo "tools/eigen_bench.sh: launch order warm-up (4 jobs), cold, [cold, warm from kappa +-1 %], [cold, warm +-10 %], [cold, warm from any kappa in [1, 10]]"; bash tools/eigen_bench.sh 2>&1 | grep "eigen launch";
  echo "tools/eig_abl.sh: 120 rounds (8 sweeps) forced; bit 1 = no regrouping, 2 = no rotation arithmetic, 4 = no stop test, 8 = no B update, 16 = no V update, R0 = no rounds at all (set-up + output; the launches after a warm start add the two 64^3 products)"; bash tools/eig_abl.sh 0 1 2 4 8 16 31 R0; } > $O/eigen_kernel.txt 2>&1
# ---- pcsf_prepare_block_scoring (stream pool warmed while the file is read) on / off, alternating on one box
bash tools/ab_prepare.sh > $O/prepare_blocks.txt 2>&1
# ---- tracks mode: all-gap subtrees skipped per work-group (tools/skip_e2e.sh)
python3 tools/gen_synthetic_maf.py /tmp/syn1mb.maf --bases 1000000 > /dev/null 2>&1
python3 tools/gen_synthetic_maf.py /tmp/clade1mb.maf --bases 1000000 --clade-dropout > $O/clade_maf.txt 2>&1
python3 tools/replicate_maf.py /tmp/clade1mb.maf /tmp/clade10mb.maf --copies 10 > /dev/null 2>&1
python3 tools/replicate_maf.py /tmp/syn1mb.maf /tmp/syn10mb.maf --copies 10 > /dev/null 2>&1
{ for f in syn10mb clade10mb; do for skip in 1 0; do echo "== $f.maf, PCSF_SKIP_ALLGAP=$skip"; for i in 1 2; do PCSF_SKIP_ALLGAP=$skip phylocsfpp_amd/host/phylocsf-amd build-tracks --output /tmp/out_$f$skip 58mammals /tmp/$f.maf 2>&1 | tail -2; done; done;
    echo "outputs of the two runs: $(diff -rq /tmp/out_${f}1 /tmp/out_${f}0 > /dev/null && echo identical || echo DIFFERENT)"; done; } > $O/skip_allgap.txt 2>&1
for f in syn10mb clade10mb; do rm -rf /tmp/pk; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -- phylocsfpp_amd/host/phylocsf-amd build-tracks --output /tmp/out_pk 58mammals /tmp/$f.maf > /dev/null 2>&1; echo "== $f.maf"; head -4 $(find /tmp/pk -name "*kernel_stats.csv" | head -1) | cut -c1-170; done > $O/skip_allgap_kernel_stats.txt 2>&1
# ---- prune_factored: cycles of a tile by content (in-kernel stamps; lone work-groups and a loaded device)
{ echo "== PCSF_FACTORED_STAMPS=1 (summary stamps only)"; python3 tools/factored_tile_cost.py 200 2>&1 | grep "stamps"; python3 tools/factored_tile_cost.py 3000 2>&1 | grep stamps; echo "== PCSF_FACTORED_STAMPS=2 (every op that runs is stamped too: ~400 cycles each)"; PCSF_FACTORED_STAMPS=2 python3 tools/factored_tile_cost.py 200 2>&1 | grep -A1 "stamps"; } > $O/factored_tile_cost.txt 2>&1
# ---- bls_kernel inside a build-tracks run (kernel statistics of the segment tiled ten times)
bash tools/bls_ab.sh $O/bls > /dev/null 2>&1; cp $O/bls/kernel_stats.txt $O/build_tracks_kernel_stats.txt; rm -rf $O/bls
# ---- the smoothing pass: device recurrences against the host pass, and one long run cut into chunks
PCSF_HMM_STATS=1 bash tools/smooth_time.sh $O/smooth > /dev/null 2>&1; cp $O/smooth/smooth_time.txt $O/smoothing_pass.txt; rm -rf $O/smooth
{ echo "== PCSF_HOST_HMM=1 (host/hmm.hpp, six threads)"; P=$(python3 -c "import json;print(','.join('%.17g'%x for x in json.load(open('tests/golden/build-tracks/hmm_params_backsolved.json'))['hmm_params']))");
  for i in 1 2; do PCSF_HOST_HMM=1 phylocsfpp_amd/host/phylocsf-amd build-tracks --output /tmp/out_smh --output-phylo 1 --output-regions 1 --hmm-params $P 58mammals /tmp/syn10mb.maf 2>&1 | grep "seconds:"; done;
  echo "smoothed files of the two passes: $(for f in /tmp/out_sm/PhyloCSF[+-]*; do cmp -s $f /tmp/out_smh/$(basename $f) || echo DIFFERENT $f; done | head -3; echo compared)"; } >> $O/smoothing_pass.txt 2>&1
{ python3 tools/hmm_long_run.py 5000000; python3 tools/hmm_long_run.py 2000000 constant; } 2>&1 | grep -v amdgpu.ids > $O/hmm_long_run.txt
# ---- BASELINE.json configs[3] on one GPU (100 Mb, text and packed)
bash tools/config4_one_gpu.sh tmp_c4 100 > /dev/null 2>&1; mkdir -p $O/config4; cp gpurun_out/tmp_c4/config4/* $O/config4/; rm -rf gpurun_out/tmp_c4
# ---- BASELINE.json configs[4] on one GPU at 20 Mb: 100vertebrates, HMM parameters estimated from the exon list, device against host smoothing
bash tools/config5_one_gpu.sh $O/c5 20 > /dev/null 2>&1; cp $O/c5/config5.txt $O/config5_one_gpu.txt; rm -rf $O/c5
ls -la $O
