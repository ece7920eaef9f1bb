#!/bin/bash
# build-tracks with / without the skipping of all-gap subtrees (PCSF_SKIP_ALLGAP) on the bench's synthetic MAF and on one with clade-level
# species drop-out (tools/gen_synthetic_maf.py --clade-dropout), ten 1 Mb segments each
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
python3 tools/gen_synthetic_maf.py /tmp/syn1mb.maf --bases 1000000 > /dev/null 2>&1
python3 tools/gen_synthetic_maf.py /tmp/clade1mb.maf --bases 1000000 --clade-dropout 2>&1 | tail -1
python3 tools/replicate_maf.py /tmp/clade1mb.maf /tmp/clade10mb.maf --copies 10 > /dev/null 2>&1
python3 tools/replicate_maf.py /tmp/syn1mb.maf /tmp/syn10mb.maf --copies 10 > /dev/null 2>&1
for f in syn10mb clade10mb; do
  for skip in 1 0; do echo "== $f.maf, PCSF_SKIP_ALLGAP=$skip"; for i in 1 2; do PCSF_SKIP_ALLGAP=$skip phylocsfpp_amd/host/phylocsf-amd build-tracks --output /tmp/out_$f$skip 58mammals /tmp/$f.maf 2>&1 | tail -2; done; done
  echo "outputs of the two runs: $(diff -rq /tmp/out_${f}1 /tmp/out_${f}0 > /dev/null && echo identical || echo DIFFERENT)"
done
for f in syn10mb clade10mb; do for skip in 1 0; do rm -rf /tmp/pk; PCSF_SKIP_ALLGAP=$skip rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -- phylocsfpp_amd/host/phylocsf-amd build-tracks --output /tmp/out_pk 58mammals /tmp/$f.maf > /dev/null 2>&1; echo "== $f.maf PCSF_SKIP_ALLGAP=$skip"; sed -n 2,2p $(find /tmp/pk -name "*kernel_stats.csv" | head -1) | cut -c1-150; done; done
