#!/bin/bash
# Kernel times of build-tracks with the files' text formatted on the device (csrc/wig.hip): a 20 Mb synthetic 58mammals file, packed input.
#   bash tools/r04_wig_profile.sh <tag>        outputs: gpurun_out/<tag>/wig_profile/
tag=${1:-r04}
out=gpurun_out/$tag/wig_profile; mkdir -p $out
work=${TMPDIR:-/tmp}/pcsf_wig; rm -rf $work; mkdir -p $work
cli=phylocsfpp_amd/host/phylocsf-amd
model=tests/golden/models/58mammals
python tools/gen_synthetic_maf.py $work/seg.maf --bases 1000000 > /dev/null 2>&1
python tools/replicate_maf.py $work/seg.maf $work/c.maf --copies 20 > /dev/null 2>&1
$cli pack-maf $model $work/c.maf $work/c.pcsfbin 2> /dev/null
for mode in 1 0; do
  for rep in 1 2; do PCSF_DEVICE_WIG=$mode $cli build-tracks --output $work/o$mode $model $work/c.pcsfbin 2> $out/run_wig$mode.txt; done
done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $work/prof -o wig -- $cli build-tracks --output $work/o1 $model $work/c.pcsfbin > $out/rocprof.log 2>&1
f=$(find $work/prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $out/kernel_stats.csv
# the same run with the host formatting the text: what the scoring kernels take without the wig kernels beside them
export PCSF_DEVICE_WIG=0
rocprofv3 --kernel-trace --stats --output-format csv -d $work/prof0 -o wig -- $cli build-tracks --output $work/o0 $model $work/c.pcsfbin > $out/rocprof_hostwig.log 2>&1
f=$(find $work/prof0 -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $out/kernel_stats_hostwig.csv
cat $out/run_wig1.txt $out/run_wig0.txt; cut -c1-150 $out/kernel_stats.csv | head -12; cut -c1-150 $out/kernel_stats_hostwig.csv | head -8
rm -rf $work
