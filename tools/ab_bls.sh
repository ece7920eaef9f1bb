#!/bin/bash
# GPU box: build-tracks end to end with the branch-length scores from the device kernel vs on the host
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; O=gpurun_out/bls; mkdir -p $O
make -s -C phylocsfpp_amd/host
python3 tools/gen_synthetic_maf.py /tmp/syn5mb.maf --bases 5000000 > /dev/null 2>&1
for h in 0 1; do
  for i in 1 2 3; do
    echo "--host-bls $h"
    phylocsfpp_amd/host/phylocsf-amd build-tracks --host-bls $h --output /tmp/out_bt$h tests/golden/models/58mammals /tmp/syn5mb.maf 2>&1 | tail -2
  done
done > $O/ab_bls.txt
for n in PhyloCSFpower PhyloCSFRaw+1 PhyloCSFRaw-3; do cmp /tmp/out_bt0/$n.wig /tmp/out_bt1/$n.wig && echo "$n identical" >> $O/ab_bls.txt; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- phylocsfpp_amd/host/phylocsf-amd build-tracks --output /tmp/out_bt2 tests/golden/models/58mammals /tmp/syn5mb.maf > /dev/null 2>&1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/build_tracks_kernel_stats.csv; rm -rf $O/prof
cat $O/ab_bls.txt; cat $O/build_tracks_kernel_stats.csv
