#!/bin/bash
# BASELINE.json configs[3] (build-tracks 58mammals on a synthetic 100 Mb MAF) on ONE GPU, end to end: the 1/8 share an
# 8-GPU node would see per GPU is this run / 8.  The 100 Mb file = 100 copies of the generator's 1 Mb segment along
# one chromosome (tools/replicate_maf.py; ~9 GB of text).  Text path and packed path (pack-maf) are timed separately,
# as SURVEY.md 8d.4 asks, and must give identical files.
#   bash tools/config4_one_gpu.sh <tag> [copies]       outputs: gpurun_out/<tag>/config4/
tag=${1:-r02}; copies=${2:-100}
out=gpurun_out/$tag/config4; mkdir -p $out
work=${TMPDIR:-/tmp}/pcsf_c4; rm -rf $work; mkdir -p $work
cli=phylocsfpp_amd/host/phylocsf-amd
model=tests/golden/models/58mammals
{ echo "host: $(nproc) hardware threads, cgroup cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; free -g | head -2; df -h $work | tail -1; } > $out/system.txt
avail_gb=$(awk '/MemAvailable/ {print int($2/1048576)}' /proc/meminfo)
disk_gb=$(df -BG --output=avail $work | tail -1 | tr -dc 0-9)
need=$((copies / 4 + 8))
if [ "$avail_gb" -lt "$need" ] || [ "$disk_gb" -lt "$need" ]; then
  echo "not enough memory ($avail_gb GB) or disk ($disk_gb GB) for $copies copies (need ~$need GB each)" | tee -a $out/system.txt; exit 0
fi
python tools/gen_synthetic_maf.py $work/seg.maf --bases 1000000 > $out/generate.txt 2>&1
python tools/replicate_maf.py $work/seg.maf $work/c4.maf --copies $copies >> $out/generate.txt 2>&1
ls -l $work/c4.maf >> $out/generate.txt
for rep in 1 2; do   # the second run reads the file from the page cache
  $cli build-tracks --output $work/out_text $model $work/c4.maf 2> $out/text_run$rep.txt
done
( time $cli pack-maf $model $work/c4.maf $work/c4.pcsfbin ) 2> $out/pack.txt
ls -l $work/c4.pcsfbin >> $out/pack.txt
# A/B between the two packed runs (run order matters on these boxes: the later run of two equal ones is the faster): the seven files'
# text formatted by the host (round 3's stage 3) instead of on the device (csrc/wig.hip)
$cli build-tracks --output $work/out_packed $model $work/c4.pcsfbin 2> $out/packed_run1.txt
PCSF_DEVICE_WIG=0 $cli build-tracks --output $work/out_hostwig $model $work/c4.pcsfbin 2> $out/packed_hostwig_run.txt
$cli build-tracks --output $work/out_packed $model $work/c4.pcsfbin 2> $out/packed_run2.txt
same=yes; same_wig=yes
for f in $work/out_text/*.wig; do
  cmp -s $f $work/out_packed/$(basename $f) || same=no
  cmp -s $f $work/out_hostwig/$(basename $f) || same_wig=no
done
{ echo "text and packed runs wrote identical files: $same"; echo "device-formatted and host-formatted text identical: $same_wig"; ls -l $work/out_text; } > $out/outputs.txt
cat $out/text_run2.txt $out/pack.txt $out/packed_run2.txt $out/packed_hostwig_run.txt $out/outputs.txt
rm -rf $work
