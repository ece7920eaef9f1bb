cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; rm -rf gpurun_out/prof_expm
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_expm -- python3 tools/expm_bench.py 4000 > gpurun_out/prof_expm.log 2>&1
grep expm_kernel $(find gpurun_out/prof_expm -name "*kernel_stats.csv" | head -1)
