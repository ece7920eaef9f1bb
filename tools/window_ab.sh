#!/bin/bash
# On the GPU box: build-tracks on the synthetic segment tiled COPIES times (default 20 = 1.8 GB of text) under 32 / 64 / 128 MB pipeline steps
# (--window-bytes): with the table form's kernel a 32 MB step is ~5.5 ms of GPU work -- do longer steps pay now?
cd "$(dirname "$0")/.." || exit 1
COPIES=${COPIES:-20}
W=${TMPDIR:-/tmp}/window_ab_$$; mkdir -p $W
make -s -C phylocsfpp_amd/host
python3 tools/gen_synthetic_maf.py $W/seg.maf --bases 1000000 > /dev/null 2>&1
python3 tools/replicate_maf.py $W/seg.maf $W/big.maf --copies $COPIES > /dev/null 2>&1
CLI=phylocsfpp_amd/host/phylocsf-amd
for rep in 1 2; do
  for wb in 0 67108864 134217728; do
    echo -n "window-bytes $wb: "; $CLI build-tracks --output $W/out --window-bytes $wb 58mammals $W/big.maf 2>&1 | grep 'seconds: total'
  done
done
rm -rf $W
