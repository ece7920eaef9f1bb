#!/bin/bash
# BASELINE.json configs[4] (build-tracks 100vertebrates + --output-phylo, HMM parameters estimated from a coding-exon list) on ONE GPU:
# the generator's 100-species 1 Mb segment (all 100 rows mapped by the model) tiled along one chromosome, the exon list tiled with it.
# The smoothing pass with the device recurrences (default) and as host code (PCSF_HOST_HMM=1), the 19 output files compared.
#   bash tools/config5_one_gpu.sh <out dir> [copies]
cd ${GRAFT_REPO_ROOT:-.}; O=${1:-gpurun_out/config5}; copies=${2:-20}; mkdir -p $O
W=${TMPDIR:-/tmp}/pcsf_c5; rm -rf $W; mkdir -p $W
make -s -C phylocsfpp_amd/host
python3 tools/gen_synthetic_maf.py $W/seg.maf --bases 1000000 --seed 20261002 --exons $W/seg.exons > $O/generate.txt 2>&1
python3 tools/replicate_maf.py $W/seg.maf $W/c5.maf --copies $copies >> $O/generate.txt 2>&1
python3 - $W/seg.exons $W/c5.exons $copies <<'P'
import sys
src, dst, copies = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows = [l.split() for l in open(src) if l.strip()]
with open(dst, "w") as f:
    for k in range(copies):
        for c, s, fr, a, b in rows:
            f.write("%s\t%s\t%s\t%d\t%d\n" % (c, s, fr, int(a) + k * 1100000, int(b) + k * 1100000))
P
G=$((copies * 1100000 + 1000000))
ls -l $W/c5.maf $W/c5.exons >> $O/generate.txt
for v in device host; do for rep in 1 2; do
  E=""; [ $v = host ] && E="PCSF_HOST_HMM=1"
  echo "== smoothing recurrences: $v (run $rep)"
  env $E PCSF_HMM_STATS=1 phylocsfpp_amd/host/phylocsf-amd build-tracks --output $W/out_$v --output-phylo 1 --output-regions 1 --genome-length $G --coding-exons $W/c5.exons 100vertebrates $W/c5.maf 2>&1 | grep -E "HMM parameters|seconds:|hmm_smooth_runs|smooth_tracks_device|codon-frames"
done; done > $O/config5.txt 2>&1
{ same=yes; n=0; for f in $W/out_host/*; do n=$((n + 1)); cmp -s $f $W/out_device/$(basename $f) || same=no; done; echo "$n files; device and host smoothing wrote identical files: $same"; ls -l $W/out_device | head -24; } >> $O/config5.txt
cat $O/config5.txt; rm -rf $W
