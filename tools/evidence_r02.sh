#!/bin/bash
# Runs on the GPU box: the measurements DESIGN.md quotes for round 2.  bash tools/evidence_r02.sh [tag]  -> gpurun_out/<tag>/
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; O=gpurun_out/${1:-evidence_r02}; mkdir -p $O
# block strategies: wall time, where a round's time goes, per-kernel totals
PCSF_TIME_BLOCKS=1 python3 tools/bench_blocks.py > $O/blocks_bench_516.txt 2>&1
PCSF_TIME_BLOCKS=1 python3 tools/bench_blocks.py - 0 4 nocpu > $O/blocks_bench_2064.txt 2>&1
PCSF_FACTORED_MAX_TILES=0 python3 tools/bench_blocks.py - 0 1 nocpu > $O/blocks_bench_516_explicit_only.txt 2>&1
PCSF_BLOCK_GROUPS=1 python3 tools/bench_blocks.py - 0 1 nocpu > $O/blocks_bench_516_one_group.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_blocks -- python3 tools/bench_blocks.py - 0 4 nocpu > /dev/null 2>&1
cp $(find $O/prof_blocks -name "*kernel_stats.csv" | head -1) $O/blocks_kernel_stats.csv
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_blocks -- python3 tools/bench_blocks.py - 0 4 nocpu > /dev/null 2>&1
python3 - <<PY
import csv, glob, json
acc = {}
for f in glob.glob("$O/pmc_blocks/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "").split("(")[0]
        acc.setdefault(k, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
json.dump({k: {c: {"sum": sum(v), "dispatches": len(v)} for c, v in d.items()} for k, d in acc.items()}, open("$O/blocks_pmc_summary.json", "w"), indent=1)
PY
rm -rf $O/prof_blocks $O/pmc_blocks
# end to end
make -s -C phylocsfpp_amd/host
python3 tools/gen_synthetic_maf.py /tmp/syn1mb.maf --bases 1000000 > /dev/null 2>&1
python3 tools/gen_synthetic_maf.py /tmp/syn5mb.maf --bases 5000000 > /dev/null 2>&1
{ for f in /tmp/syn1mb.maf /tmp/syn5mb.maf; do for i in 1 2 3; do phylocsfpp_amd/host/phylocsf-amd build-tracks --output /tmp/out_bt tests/golden/models/58mammals $f 2>&1 | tail -2; done; done; } > $O/build_tracks_end_to_end.txt
{ echo "== separate extract_frames_kernel (PCSF_FUSED_FRONTEND=0)"; for i in 1 2 3; do PCSF_FUSED_FRONTEND=0 phylocsfpp_amd/host/phylocsf-amd build-tracks --output /tmp/out_bt tests/golden/models/58mammals /tmp/syn5mb.maf 2>&1 | tail -1; done; } >> $O/build_tracks_end_to_end.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bt -- phylocsfpp_amd/host/phylocsf-amd build-tracks --output /tmp/out_bt tests/golden/models/58mammals /tmp/syn5mb.maf > /dev/null 2>&1
cp $(find $O/prof_bt -name "*kernel_stats.csv" | head -1) $O/build_tracks_kernel_stats.csv
rm -rf $O/prof_bt
zcat tests/golden/score-msa/chr22.516alignments.maf.gz > /tmp/s516.maf; { for s in fixed mle omega; do a=$(date +%s.%N); phylocsfpp_amd/host/phylocsf-amd score-msa --strategy $s tests/golden/models/100vertebrates /tmp/s516.maf > /dev/null 2>&1; b=$(date +%s.%N); python3 -c "print(\"score-msa --strategy $s, 516 blocks: %.2f s wall (process start to exit)\" % ($b - $a))"; done; } > $O/score_msa_cli.txt 2>&1 || true
ls -la $O
# batch size: the same 10 Mb (ten 1 Mb segments) with 32 MB (default) and 64 MB windows of MAF text per pipeline step
python3 tools/replicate_maf.py /tmp/syn1mb.maf /tmp/syn10mb.maf --copies 10 > /dev/null 2>&1
{ for w in 33554432 67108864 134217728; do echo "== --window-bytes $w"; for i in 1 2 3; do phylocsfpp_amd/host/phylocsf-amd build-tracks --window-bytes $w --output /tmp/out_bt tests/golden/models/58mammals /tmp/syn10mb.maf 2>&1 | tail -2; done; done; } > $O/build_tracks_window.txt
python3 tools/diag_determinism.py 40 > $O/determinism.txt 2>&1
python3 tools/objective_noise.py > $O/objective_noise.txt 2>&1
python3 tools/omega_divergence.py 14 > $O/omega_divergence.txt 2>&1
ls -la $O
# ---- added at the end of round 2 (each line is what produced the file of that name under profiles/r02/) --------------------
bash tools/startup_profile.sh > $O/startup_profile.txt 2>&1
{ echo "register-resident form (default):"; bash tools/eigen_bench.sh 2>&1 | grep "eigen launch"; echo "LDS form (PCSF_EIGEN_LDS=1):"; PCSF_EIGEN_LDS=1 bash tools/eigen_bench.sh 2>&1 | grep "eigen launch"; PCSF_EIGEN_STAMPS=1 python3 tools/eigen_bench.py 200 2>&1 | grep stamps | head -4; } > $O/eigen_forms.txt
for r in 1 4 16; do { echo "== bench_blocks.py - 0 $r nocpu, register-resident eigen-solver (default)"; PCSF_TIME_BLOCKS=1 python3 tools/bench_blocks.py - 0 $r nocpu 2>&1 | grep -v amdgpu; echo "== PCSF_EIGEN_LDS=1"; PCSF_EIGEN_LDS=1 PCSF_TIME_BLOCKS=1 python3 tools/bench_blocks.py - 0 $r nocpu 2>&1 | grep "run_omega\|OMEGA"; }; done > $O/blocks_bench_eigen_forms.txt
for r in 1 2 4 8 16; do for g in 1 2; do echo "copies $r groups $g"; PCSF_BLOCK_GROUPS=$g python3 tools/bench_blocks.py - 0 $r nocpu 2>&1 | grep "GPU MLE\|GPU OMEGA"; done; done > $O/blocks_groups_raw.txt
for r in 1 4 16; do python3 tools/bench_blocks.py - 0 $r nocpu 2>&1 | grep "GPU "; done > $O/blocks_final.txt
{ python3 tools/factored_tile_cost.py 200; python3 tools/factored_tile_cost.py 3000; } 2>&1 | grep -v amdgpu > $O/factored_tile_cost.txt
for r in 1 16; do rm -rf /tmp/pb; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -- python3 tools/bench_blocks.py - 0 $r nocpu > /tmp/pb.log 2>&1; echo "== bench_blocks.py - 0 $r nocpu"; grep "GPU " /tmp/pb.log; head -8 $(find /tmp/pb -name "*kernel_stats.csv" | head -1) | cut -c1-150; done > $O/blocks_kernel_stats_final.txt
bash tools/config4_one_gpu.sh ${1:-evidence_r02}
# (profiles/r02/fuzz_against_reference.txt comes from tools/fuzz_against_reference.py in the build container, where /root/reference exists)
