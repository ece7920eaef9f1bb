#!/bin/bash
# SQ counters of eigen_omega_kernel alone (tools/eigen_bench.py, 200 jobs), per form: where a wave's cycles go -- issuing (ACTIVE_INST), parked at
# s_waitcnt / s_barrier (WAIT_ANY), stalled at issue (WAIT_INST_ANY) -- and how many instructions it issues.  Counters only (no trace options).
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
for form in ${FORMS:-quad reg lds}; do   # quad = PCSF_EIGEN_FORM=2 (seven waves, the default), reg = 1 (two waves), lds = 0
    rm -rf /tmp/prof_ep
    PCSF_EIGEN_FORM=$([ "$form" = lds ] && echo 0 || { [ "$form" = reg ] && echo 1 || echo 2; }) rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES --output-format csv -d /tmp/prof_ep -- python3 tools/eigen_bench.py ${EIGEN_JOBS:-200} > /dev/null 2>&1
    python3 - $form <<'PY'
import csv, glob, sys, collections
f = glob.glob('/tmp/prof_ep/**/*counter_collection.csv', recursive=True)[0]
per = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if 'eigen_omega' not in r['Kernel_Name']: continue
    per.setdefault(r['Dispatch_Id'], {})[r['Counter_Name']] = float(r['Counter_Value'])
print('== form %s (launches in order: warm-up, cold, [cold, warm 1 %%], [cold, warm 10 %%], [cold, warm anywhere])' % sys.argv[1])
for d, c in per.items():
    w = max(c.get('SQ_WAVES', 1), 1)
    print('  waves %5d | per wave: cycles(x4) %9.0f  active %9.0f  wait_any %9.0f  wait_inst %9.0f | VALU %8.0f  LDS %7.0f  SALU %7.0f' % (
        w, c.get('SQ_WAVE_CYCLES', 0) / w, c.get('SQ_ACTIVE_INST_ANY', 0) / w, c.get('SQ_WAIT_ANY', 0) / w, c.get('SQ_WAIT_INST_ANY', 0) / w,
        c.get('SQ_INSTS_VALU', 0) / w, c.get('SQ_INSTS_LDS', 0) / w, c.get('SQ_INSTS_SALU', 0) / w))
PY
done
