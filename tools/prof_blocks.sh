cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_blocks -- python3 tools/bench_blocks.py tests/golden/score-msa/chr22.516alignments.maf.gz 0 4 > gpurun_out/prof_blocks.log 2>&1
cat $(find gpurun_out/prof_blocks -name "*kernel_stats.csv" | head -1)
grep GPU gpurun_out/prof_blocks.log
