#!/bin/bash
# GPU box: build-tracks end to end for several pipeline window sizes (MAF bytes per stage-1 step = one backend call here)
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; O=gpurun_out/window; mkdir -p $O
make -s -C phylocsfpp_amd/host
python3 tools/gen_synthetic_maf.py /tmp/syn5mb.maf --bases 5000000 > /dev/null 2>&1
python3 tools/gen_synthetic_maf.py /tmp/syn20mb.maf --bases 20000000 > /dev/null 2>&1
for f in /tmp/syn5mb.maf /tmp/syn20mb.maf; do
for w in 16 32 64 128; do
  for i in 1 2; do
    echo "$f --window-bytes ${w}MB"
    phylocsfpp_amd/host/phylocsf-amd build-tracks --window-bytes $((w<<20)) --output /tmp/out_w tests/golden/models/58mammals $f 2>&1 | tail -1
  done
done; done 2>&1 | tee $O/ab_window.txt
