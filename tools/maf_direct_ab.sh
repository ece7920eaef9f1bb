#!/bin/bash
# Stage 1 of build-tracks on text input, alone, on the host cores of the box: tools/_build/maf_direct_bench (this tree's reader) and, when
# present, tools/_build/maf_direct_bench_r04 (round 4's) on the synthetic 1 Mb segment tiled ten times (900 MB), 16 and 8 threads.
cd "$(dirname "$0")/.." || exit 1
W=${TMPDIR:-/tmp}/maf_ab_$$
mkdir -p $W
python3 - $W <<'PY'
import sys
sys.path.insert(0, "tools")
import gen_synthetic_maf as gen, replicate_maf, contextlib, io
w = sys.argv[1]
gen.generate(w + "/seg.maf", bases=1_000_000, seed=20261002)
with contextlib.redirect_stdout(io.StringIO()):
    replicate_maf.tile(w + "/seg.maf", w + "/seg10.maf", 10, 1_100_000)
PY
nproc
for b in maf_direct_bench_r04 maf_direct_bench; do
    [ -x tools/_build/$b ] || continue
    echo "== $b"
    for t in ${THREADS:-16 16 8}; do tools/_build/$b tests/golden/models/58mammals $W/seg10.maf $t; done
done
rm -rf $W
