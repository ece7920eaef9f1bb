#!/bin/bash
# On the GPU box: kernel statistics of `phylocsf-amd build-tracks` on the synthetic segment tiled ten times (rocprofv3 --kernel-trace --stats):
# what the six-frame entry point's kernels (fused front end + all-gap skipping + subtree tables) take per codon-frame beside the bench's raw form.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
W=$TMPDIR/cli_prof_$$; mkdir -p $W
make -s -C phylocsfpp_amd/host
python3 tools/gen_synthetic_maf.py $W/seg.maf --bases 1000000 > /dev/null 2>&1
python3 tools/replicate_maf.py $W/seg.maf $W/big.maf --copies 10 > /dev/null 2>&1
export PCSF_FULL_TEARDOWN=1
for form in default full; do
  [ $form = full ] && export PCSF_CHERRY_TABLES=0 || unset PCSF_CHERRY_TABLES
  rocprofv3 --kernel-trace --stats --output-format csv -d $W/prof_$form -- phylocsfpp_amd/host/phylocsf-amd build-tracks --output $W/out 58mammals $W/big.maf > $W/log_$form.txt 2>&1
  echo "== tracks mode: $form"; grep 'codon-frames\|seconds: total' $W/log_$form.txt
  python3 - $W/prof_$form <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) >= 0.3:
            print("   %-100s calls %5s  total %9.3f ms  avg %9.1f us  %5.1f %%" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
done
rm -rf $W
