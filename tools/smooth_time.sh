#!/bin/bash
# Runs on the GPU box: what --output-phylo 1 --output-regions 1 (the smoothing pass) adds to a build-tracks run of the synthetic segment
# tiled ten times.  bash tools/smooth_time.sh <out dir>
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; O=${1:-gpurun_out/smooth}; mkdir -p $O
make -s -C phylocsfpp_amd/host
python3 tools/gen_synthetic_maf.py /tmp/syn1mb.maf --bases 1000000 > /dev/null 2>&1
python3 tools/replicate_maf.py /tmp/syn1mb.maf /tmp/syn10mb.maf --copies 10 > /dev/null 2>&1
P=$(python3 -c "import json;print(','.join('%.17g'%x for x in json.load(open('tests/golden/build-tracks/hmm_params_backsolved.json'))['hmm_params']))")
for i in 1 2; do
  echo "== raw tracks only"; phylocsfpp_amd/host/phylocsf-amd build-tracks --output /tmp/out_raw 58mammals /tmp/syn10mb.maf 2>&1 | tail -2
  echo "== --output-phylo 1 --output-regions 1"; phylocsfpp_amd/host/phylocsf-amd build-tracks --output /tmp/out_sm --output-phylo 1 --output-regions 1 --hmm-params $P 58mammals /tmp/syn10mb.maf 2>&1 | grep -E "seconds:|hmm_smooth_runs|smooth_tracks_device"
done > $O/smooth_time.txt 2>&1
( cd /tmp/out_sm && md5sum *.wig *.bed ) > $O/md5.txt 2>&1
cat $O/smooth_time.txt
