#!/bin/bash
# PMC counters of score-msa on the configs[2] MAF, summed per kernel: tools/c3_pmc.sh <strategy> <out json> (env passed through)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
W=/tmp/config3_work; mkdir -p $W/outp
MAF=$W/syn1mb.maf
[ -f $MAF ] || python3 tools/gen_synthetic_maf.py $MAF --bases 1000000 --seed 20261002 > /dev/null 2>&1
rm -rf /tmp/pmc_c3
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_c3 -- phylocsfpp_amd/host/phylocsf-amd score-msa --strategy $1 --output $W/outp tests/golden/models/58mammals $MAF > /dev/null 2>&1
mkdir -p $(dirname $2)
python3 - $2 <<'PY'
import csv, glob, json, sys
acc = {}
for f in glob.glob("/tmp/pmc_c3/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "").split("(")[0]
        acc.setdefault(k, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
out = {k: {c: {"sum": sum(v), "dispatches": len(v)} for c, v in d.items()} for k, d in acc.items()}
json.dump(out, open(sys.argv[1], "w"), indent=1)
for k, d in out.items():
    if "SQ_INSTS_MFMA" in d:
        g = d["GRBM_GUI_ACTIVE"]["sum"] / 8.0   # summed over the 8 XCDs
        print("%-60s dispatches %5d  MFMA %.3g  VALU %.3g  LDS %.3g  bank-conflict cycles %.3g  MFMA busy %.3g  GUI-active/8 %.3g -> MFMA busy / (1024 SIMDs x active cycles) = %.3f" % (
            k[:60], d["SQ_INSTS_MFMA"]["dispatches"], d["SQ_INSTS_MFMA"]["sum"], d["SQ_INSTS_VALU"]["sum"], d["SQ_INSTS_LDS"]["sum"],
            d["SQ_LDS_BANK_CONFLICT"]["sum"], d["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"], g, d["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"] / max(1.0, 1024.0 * g)))
PY
