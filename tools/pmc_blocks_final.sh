#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp; O=gpurun_out/pmc_final; mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_blocks -- python3 tools/bench_blocks.py - 0 4 nocpu > /dev/null 2>&1
python3 - <<PY
import csv, glob, json
acc = {}
for f in glob.glob("/tmp/pmc_blocks/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "").split("(")[0]
        acc.setdefault(k, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
json.dump({k: {c: {"sum": sum(v), "dispatches": len(v)} for c, v in d.items()} for k, d in acc.items()}, open("$O/blocks_pmc_summary_final.json", "w"), indent=1)
PY
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_blocks -- python3 tools/bench_blocks.py - 0 4 nocpu > /dev/null 2>&1
cp $(find /tmp/prof_blocks -name "*kernel_stats.csv" | head -1) $O/blocks_kernel_stats_2064_final.csv
head -7 $O/blocks_kernel_stats_2064_final.csv | cut -c1-160
