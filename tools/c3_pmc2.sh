#!/bin/bash
# wave-state PMC counters of score-msa on the configs[2] MAF: tools/c3_pmc2.sh <strategy> [kernel substring]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
W=/tmp/config3_work; mkdir -p $W/outp
MAF=$W/syn1mb.maf
[ -f $MAF ] || python3 tools/gen_synthetic_maf.py $MAF --bases 1000000 --seed 20261002 > /dev/null 2>&1
rm -rf /tmp/pmc2_c3
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/pmc2_c3 -- phylocsfpp_amd/host/phylocsf-amd score-msa --strategy $1 --output $W/outp tests/golden/models/58mammals $MAF > /dev/null 2>&1
python3 - ${2:-prune_tiles} <<'PY'
import csv, glob, sys
acc = {}
for f in glob.glob("/tmp/pmc2_c3/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if sys.argv[1] in row.get("Kernel_Name", ""):
            acc[row["Counter_Name"]] = acc.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
wc = acc.get("SQ_WAVE_CYCLES", 1.0)
for k, v in sorted(acc.items()):
    print("%-24s %.4g  (%.3f of SQ_WAVE_CYCLES)" % (k, v, v / wc))
PY
