#!/bin/bash
# Round 5: OMEGA on BASELINE.json configs[2] (synthetic 1 Mb MAF, 58mammals, 5,088 blocks) under the eigen-solver's two register-resident
# forms, alternating on one box, and the kernel statistics of the run with the default form.  tools/r05_omega_forms.sh [out dir]
cd "$(dirname "$0")/.." || exit 1
O=${1:-gpurun_out/r05_omega_forms}
mkdir -p "$O"
export TMPDIR=/tmp
W=/tmp/r05_work
mkdir -p $W/out
MAF=$W/syn1mb.maf
make -s -C phylocsfpp_amd/host || exit 1
[ -f $MAF ] || python3 tools/gen_synthetic_maf.py $MAF --bases 1000000 --seed 20261002 > /dev/null 2>&1 || exit 1
CLI=phylocsfpp_amd/host/phylocsf-amd
{
for rep in 1 2 3; do
  for f in 1 2; do
    echo "== configs[2] score-msa --strategy omega, PCSF_EIGEN_FORM=$f, run $rep: $(PCSF_EIGEN_FORM=$f PCSF_TIME_BLOCKS=1 $CLI score-msa --strategy omega --comp-anc 0 --output $W/out tests/golden/models/58mammals $MAF 2>&1 | grep "scoring backend")"
  done
done
} > "$O/config3_omega_forms.txt" 2>&1
cat "$O/config3_omega_forms.txt"
rm -rf /tmp/prof_r05o
PCSF_FULL_TEARDOWN=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r05o -- $CLI score-msa --strategy omega --comp-anc 0 --output $W/out tests/golden/models/58mammals $MAF > /dev/null 2>&1
cp $(find /tmp/prof_r05o -name "*kernel_stats.csv" | head -1) "$O/config3_omega_kernel_stats.csv" && head -10 "$O/config3_omega_kernel_stats.csv" | cut -c1-160
rm -rf /tmp/prof_r05o
PCSF_FULL_TEARDOWN=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r05o -- python3 tools/bench_blocks.py - 0 1 nocpu > /dev/null 2>&1
cp $(find /tmp/prof_r05o -name "*kernel_stats.csv" | head -1) "$O/blocks516_kernel_stats.csv" && head -12 "$O/blocks516_kernel_stats.csv" | cut -c1-160
