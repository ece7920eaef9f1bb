#!/bin/bash
# GPU box: timeline (kernels + memory copies, no counters) of build-tracks on the 5 Mb synthetic alignment
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; O=gpurun_out/timeline; rm -rf $O; mkdir -p $O
make -s -C phylocsfpp_amd/host
python3 tools/gen_synthetic_maf.py /tmp/syn5mb.maf --bases 5000000 > /dev/null 2>&1
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/prof -- phylocsfpp_amd/host/phylocsf-amd build-tracks --output /tmp/out_tl tests/golden/models/58mammals /tmp/syn5mb.maf > $O/run.txt 2>&1
find $O/prof -name "*.csv" | head
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob('gpurun_out/timeline/prof/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K ' + r['Kernel_Name'][:40]))
for f in glob.glob('gpurun_out/timeline/prof/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C ' + r.get('Direction', r.get('Name', '?'))))
rows.sort()
t0 = rows[0][0]
out = open('gpurun_out/timeline/timeline.txt', 'w')
prev_end = t0
for s, e, n in rows:
    out.write("%10.3f ms  +%8.3f  dur %8.3f  %s\n" % ((s - t0) / 1e6, (s - prev_end) / 1e6, (e - s) / 1e6, n))
    prev_end = max(prev_end, e)
out.close()
PY
wc -l $O/timeline.txt
