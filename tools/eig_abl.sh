#!/bin/bash
# eigen_omega_kernel<1> (PCSF_EIGEN_FORM=1, the two-wave form of rounds 2-4) with parts switched off (timing only; 120 rounds = 8 sweeps forced): tools/eig_abl.sh
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for v in "$@"; do
  rm -rf /tmp/prof_ea
  PCSF_EIGEN_FORM=1 PCSF_LIBRARY=$PWD/tools/_build/lib_eig$v.so rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ea -- python3 tools/eigen_bench.py ${EIG_JOBS:-200} > /dev/null 2>&1
  python3 - $v <<'PY'
import csv, glob, sys
f = glob.glob('/tmp/prof_ea/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'eigen_omega' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
print('variant %s:' % sys.argv[1], ' '.join('%.3f' % ((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6) for r in rows), 'ms')
PY
done
