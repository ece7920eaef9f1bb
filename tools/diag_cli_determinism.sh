#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
python tools/diag_determinism.py 20
cli=phylocsfpp_amd/host/phylocsf-amd; model=tests/golden/models/58mammals; w=/tmp/dg; rm -rf $w; mkdir -p $w
python tools/gen_synthetic_maf.py $w/seg.maf --bases 1000000 > /dev/null
python tools/replicate_maf.py $w/seg.maf $w/c.maf --copies 4 > /dev/null
for i in 1 2 3; do $cli build-tracks --contexts-per-device 1 --output $w/one$i $model $w/c.maf 2>/dev/null; done
for i in 1 2 3; do $cli build-tracks --output $w/two$i $model $w/c.maf 2>/dev/null; done
for i in 1 2 3; do $cli build-tracks --threads 1 --output $w/thr$i $model $w/c.maf 2>/dev/null; done
for d in one2 one3 two1 two2 two3 thr1 thr2 thr3; do n=0; for f in $w/one1/*.wig; do b=$(basename $f); k=$(diff $f $w/$d/$b | grep -c "^<"); n=$((n+k)); done; echo "one1 vs $d: $n differing lines"; done
