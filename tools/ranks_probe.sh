#!/bin/bash
# On the GPU box: the one-process-per-GPU command line (--ranks 1 --gather-root-text 1) on the synthetic 1 Mb segment tiled ten times, against
# the plain run -- where the rank's time goes (communicator set-up beside or before the scoring, RCCL's own start-up options).
cd "$(dirname "$0")/.." || exit 1
W=${TMPDIR:-/tmp}/ranks_probe_$$
mkdir -p $W
make -s -C phylocsfpp_amd/host
python3 - $W <<'PY'
import sys
sys.path.insert(0, "tools")
import gen_synthetic_maf as gen, replicate_maf, contextlib, io
w = sys.argv[1]
gen.generate(w + "/seg.maf", bases=1_000_000, seed=20261002)
with contextlib.redirect_stdout(io.StringIO()):
    replicate_maf.tile(w + "/seg.maf", w + "/seg10.maf", 10, 1_100_000)
PY
CLI=phylocsfpp_amd/host/phylocsf-amd
run() { local tag=$1; shift; echo "== $tag"; local t0=$(date +%s.%N); "$@" 2> $W/err.txt; local rc=$?; local t1=$(date +%s.%N); grep -E 'seconds: total|gather' $W/err.txt; echo "   rc=$rc process wall $(python3 -c "print('%.3f' % ($t1 - $t0))") s"; }
for rep in 1 2; do
run "plain" $CLI build-tracks --output $W/plain 58mammals $W/seg10.maf
run "ranks 1, root text through the gather (communicator set up beside the scoring)" $CLI build-tracks --output $W/r1 --ranks 1 --gather-root-text 1 58mammals $W/seg10.maf
PCSF_COMM_INIT_SYNC=1 run "ranks 1, communicator set up BEFORE the scoring" $CLI build-tracks --output $W/r1s --ranks 1 --gather-root-text 1 58mammals $W/seg10.maf
done
for f in $W/plain/*; do cmp $f $W/r1/$(basename $f) || echo "DIFFERS: $f"; done
export PCSF_COMM_INIT_SYNC=1
RCCL_MSCCL_ENABLE=0 RCCL_MSCCLPP_ENABLE=0 run "before the scoring, RCCL_MSCCL_ENABLE=0 RCCL_MSCCLPP_ENABLE=0" $CLI build-tracks --output $W/r1s --ranks 1 --gather-root-text 1 58mammals $W/seg10.maf
RCCL_MSCCL_ENABLE=0 RCCL_MSCCLPP_ENABLE=0 NCCL_MAX_NCHANNELS=4 run "... and NCCL_MAX_NCHANNELS=4" $CLI build-tracks --output $W/r1s --ranks 1 --gather-root-text 1 58mammals $W/seg10.maf
NCCL_MAX_NCHANNELS=2 NCCL_MIN_NCHANNELS=2 run "NCCL_MAX_NCHANNELS=2 alone" $CLI build-tracks --output $W/r1s --ranks 1 --gather-root-text 1 58mammals $W/seg10.maf
NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT $CLI build-tracks --output $W/r1d --ranks 1 --gather-root-text 1 58mammals $W/seg.maf 2>&1 | grep -v 'WARN Could not read node\|^$\|^  seconds' | head -120
rm -rf $W
