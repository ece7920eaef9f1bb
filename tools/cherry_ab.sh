#!/bin/bash
# On the GPU box: the bench's kernel line (58mammals + the two other models) for the forms of tracks mode -- triples + cherries as leaves
# (default), cherries only (PCSF_TABLE_LEAVES=2), the full program (PCSF_CHERRY_TABLES=0) -- and for variant libraries given as arguments
# (tools/build_variant.sh; e.g. lib_rowq0.so).
cd "$(dirname "$0")/.." || exit 1
line() { python3 bench.py --steps 10 --warmup 3 --end-to-end-runs 0 --block-runs 0 --cpu-baseline-sites 0 --other-models 1 --table-forms 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; c=r.get('cherry_tables',{})
print('%-44s %7.2f M codon-frames/s  %6.3f ms  frac %.3f  products/model %s  | ' % ('$1', d['value']/1e6, d['ms_per_step'], r['frac'], c.get('products_per_model','full')), {k:(round(v['codon_frames_per_s']/1e6,1),round(v['roofline']['frac'],3)) for k,v in d.get('other_models',{}).items()})"; }
line "triples + cherries (default)"
PCSF_TABLE_LEAVES=2 line "cherries only (PCSF_TABLE_LEAVES=2)"
PCSF_CHERRY_TABLES=0 line "full program (PCSF_CHERRY_TABLES=0)"
for lib in "$@"; do PCSF_LIBRARY=$PWD/$lib line "$lib"; done
