#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace + PMC passes of bench.py.
# Usage: bash tools/profile_gpu.sh <tag>     -> gpurun_out/prof_<tag>/...
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
BENCH="python3 bench.py --steps 5 --warmup 1 --cpu-baseline-sites 0 --strong-steps 0 --other-models 0 --table-forms 0"   # the 58mammals launches only: the statistics are per kernel NAME
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU --output-format csv -d $OUT/pmc_inst -- $BENCH > $OUT/pmc_inst.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- $BENCH > $OUT/pmc_l2.log 2>&1
find $OUT -name "*.csv" | head -50 > $OUT/files.txt
# keep only small summaries (kernel_stats + the prune_fixed rows of counter collection)
python3 - <<PY
import csv, glob, os, json
out = "$OUT"
summ = {}
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    acc = {}
    for row in csv.DictReader(open(f)):
        if "prune_fixed<4, false" not in row.get("Kernel_Name", ""): continue   # the bench's launches (the one-work-group launches in which a context records its all-gap messages are the stamped variant)
        acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    for k, v in acc.items():
        summ[k] = {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)}
json.dump(summ, open(out + "/pmc_summary.json", "w"), indent=1)
if "FETCH_SIZE" in summ and "WRITE_SIZE" in summ:   # per-launch memory-side bytes the way MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE in KB, doubled)
    f, w = summ["FETCH_SIZE"]["mean_per_dispatch"], summ["WRITE_SIZE"]["mean_per_dispatch"]
    json.dump({"sites_per_launch": 2000000, "hbm_bytes_per_launch": int(f * 1024 * 2 + w * 1024), "fetch_size_kb_raw": f, "write_size_kb_raw": w,
               "dispatches": summ["FETCH_SIZE"]["dispatches"], "source": "profiles/$TAG/pmc_summary.json",
               "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over the bench command of tools/profile_gpu.sh (python3 bench.py --steps 5 --warmup 1 "
                       "--cpu-baseline-sites 0 --strong-steps 0 --other-models 0); means over the bench's own prune_fixed launches; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for "
                       "gfx950.  WRITE_SIZE is the 16 MB of scores.  FETCH_SIZE counts L2 misses including Infinity-Cache hits: re-reads of the branch-matrix fragments, "
                       "the leaves' rows and the subtree tables' rows by 31,250 work-groups; algorithmic HBM bytes are 132 MB."},
              open(out + "/traffic.json", "w"), indent=1)
print(json.dumps(summ, indent=1))
for f in glob.glob(out + "/**/*kernel_stats.csv", recursive=True):
    print(open(f).read())
PY
du -sh $OUT
