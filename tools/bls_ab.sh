#!/bin/bash
# Runs on the GPU box: bls_kernel after a change -- the tests that hold it to the reference's recursion, and its time inside a build-tracks run
# (rocprofv3 kernel statistics of the synthetic segment tiled ten times).  bash tools/bls_ab.sh <out dir>
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; O=${1:-gpurun_out/bls_ab}; mkdir -p $O
make -s -C phylocsfpp_amd/host
timeout 900 python3 -m pytest tests/test_gpu_frontend.py tests/test_gpu_cli.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
python3 tools/gen_synthetic_maf.py /tmp/syn1mb.maf --bases 1000000 > /dev/null 2>&1
python3 tools/replicate_maf.py /tmp/syn1mb.maf /tmp/syn10mb.maf --copies 10 > /dev/null 2>&1
for i in 1 2; do phylocsfpp_amd/host/phylocsf-amd build-tracks --output /tmp/out_bt 58mammals /tmp/syn10mb.maf 2>&1 | tail -2; done > $O/build_tracks.txt 2>&1
rm -rf /tmp/pk; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -- phylocsfpp_amd/host/phylocsf-amd build-tracks --output /tmp/out_pk 58mammals /tmp/syn10mb.maf > /dev/null 2>&1
head -6 $(find /tmp/pk -name "*kernel_stats.csv" | head -1) | cut -c1-200 > $O/kernel_stats.txt
cat $O/build_tracks.txt $O/kernel_stats.txt
