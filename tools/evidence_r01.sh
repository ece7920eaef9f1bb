#!/bin/bash
# Runs on the GPU box: the secondary measurements quoted in DESIGN.md (block strategies, expm kernel, end-to-end CLI).
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; O=gpurun_out/evidence; mkdir -p $O
python3 tools/bench_blocks.py > $O/blocks_bench_516.txt 2>&1
python3 tools/bench_blocks.py tests/golden/score-msa/chr22.516alignments.maf.gz 0 4 2>&1 | grep -v "CPU oracle" > $O/blocks_bench_2064.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_blocks -- python3 tools/bench_blocks.py tests/golden/score-msa/chr22.516alignments.maf.gz 0 4 > /dev/null 2>&1
cp $(find $O/prof_blocks -name "*kernel_stats.csv" | head -1) $O/blocks_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_expm -- python3 tools/expm_bench.py 4000 > $O/expm_bench.txt 2>&1
cp $(find $O/prof_expm -name "*kernel_stats.csv" | head -1) $O/expm_kernel_stats.csv
rm -rf $O/prof_blocks $O/prof_expm
make -s -C phylocsfpp_amd/host
python3 tools/gen_synthetic_maf.py /tmp/syn1mb.maf --bases 1000000 > /dev/null 2>&1
python3 tools/gen_synthetic_maf.py /tmp/syn5mb.maf --bases 5000000 > /dev/null 2>&1
{ for f in /tmp/syn1mb.maf /tmp/syn5mb.maf; do for i in 1 2 3; do phylocsfpp_amd/host/phylocsf-amd build-tracks --output /tmp/out_bt tests/golden/models/58mammals $f 2>&1 | tail -2; done; done; } > $O/build_tracks_end_to_end.txt
zcat tests/golden/score-msa/chr22.516alignments.maf.gz > /tmp/s516.maf; { for s in fixed mle omega; do a=$(date +%s.%N); phylocsfpp_amd/host/phylocsf-amd score-msa --strategy $s tests/golden/models/100vertebrates /tmp/s516.maf > /dev/null 2>&1; b=$(date +%s.%N); python3 -c "print(\"score-msa --strategy $s, 516 blocks: %.2f s wall (process start to exit)\" % ($b - $a))"; done; } > $O/score_msa_cli.txt 2>&1 || true
ls -la $O
