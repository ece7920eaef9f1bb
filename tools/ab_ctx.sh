#!/bin/bash
# GPU box: build-tracks with one and with two scoring contexts on the same GPU (transfers of one overlap kernels of the other)
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; O=gpurun_out/ctx; mkdir -p $O
make -s -C phylocsfpp_amd/host
python3 tools/gen_synthetic_maf.py /tmp/syn5mb.maf --bases 5000000 > /dev/null 2>&1
for d in 0 0,0 0,0,0; do
  for i in 1 2 3; do
    echo "--devices $d"
    phylocsfpp_amd/host/phylocsf-amd build-tracks --devices $d --output /tmp/out_c$i tests/golden/models/58mammals /tmp/syn5mb.maf 2>&1 | tail -1
  done
done > $O/ab_ctx.txt
cat $O/ab_ctx.txt
