#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
rm -rf /tmp/prof_eb
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_eb -- python3 tools/eigen_bench.py 200 2>&1 | tail -2
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_eb/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'eigen_omega' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
for r in rows:
    print('eigen launch: %.3f ms' % ((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6))
PY
