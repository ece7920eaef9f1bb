#!/bin/bash
# GPU box: build-tracks end to end for several host thread counts
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; O=gpurun_out/threads; mkdir -p $O
make -s -C phylocsfpp_amd/host
python3 tools/gen_synthetic_maf.py /tmp/syn5mb.maf --bases 5000000 > /dev/null 2>&1
cat /sys/fs/cgroup/cpu.max; cat /sys/fs/cgroup/cpu.stat | head -8
for t in 4 8 12 16 24; do
  for i in 1 2; do
    echo "--threads $t"
    phylocsfpp_amd/host/phylocsf-amd build-tracks --threads $t --output /tmp/out_t tests/golden/models/58mammals /tmp/syn5mb.maf 2>&1 | tail -1
  done
done 2>&1 | tee $O/ab_threads.txt
cat /sys/fs/cgroup/cpu.stat | head -8
