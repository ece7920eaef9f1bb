#!/bin/bash
# GPU box: build-tracks with one and with two scoring contexts on the same GPU (the kernels of the two take turns,
# the transfers of one overlap the kernels of the other), 5 Mb and 20 Mb synthetic alignments
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; O=gpurun_out/ctx; mkdir -p $O
make -s -C phylocsfpp_amd/host
python3 tools/gen_synthetic_maf.py /tmp/syn5mb.maf --bases 5000000 > /dev/null 2>&1
python3 tools/gen_synthetic_maf.py /tmp/syn20mb.maf --bases 20000000 > /dev/null 2>&1
for f in /tmp/syn5mb.maf /tmp/syn20mb.maf; do
for d in 1 2; do
  for i in 1 2 3; do
    echo "$f --contexts-per-device $d"
    phylocsfpp_amd/host/phylocsf-amd build-tracks --contexts-per-device $d --output /tmp/out_c$d tests/golden/models/58mammals $f 2>&1 | tail -1
  done
done; done > $O/ab_ctx.txt
for n in PhyloCSFpower PhyloCSFRaw+1 PhyloCSFRaw-2; do cmp /tmp/out_c1/$n.wig /tmp/out_c2/$n.wig && echo "$n identical" >> $O/ab_ctx.txt; done
cat $O/ab_ctx.txt
