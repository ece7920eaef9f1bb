#!/bin/bash
# eigen_omega_kernel alone (tools/eigen_bench.py under rocprofv3 --kernel-trace), one line per form of the kernel: launch durations in
# launch order -- warm-up, N cold, [cold, warm 1 %], [cold, warm 10 %], [cold, warm anywhere] -- and job 0's sweeps / steps / cycles
# (PCSF_EIGEN_STAMPS).  FORMS="reg lds quad" (reg = PCSF_EIGEN_FORM=1, two waves; lds = PCSF_EIGEN_LDS=1; quad = PCSF_EIGEN_FORM=2, seven waves, the default), EIGEN_JOBS=200; extra arguments:
# other builds of the library (tools/build_variant.sh), run with the default form.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
N=${EIGEN_JOBS:-200}
run() {
    rm -rf /tmp/prof_eb
    rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_eb -- python3 tools/eigen_bench.py $N > /tmp/prof_eb.log 2>&1
    grep "eigen stamps" /tmp/prof_eb.log | sed -n '2,3p;5p' | sed 's/; per step.*//; s/^/  /'
    python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_eb/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'eigen_omega' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
print('  eigen launches (ms): ' + ' '.join('%.3f' % ((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6) for r in rows))
PY
}
export PCSF_EIGEN_STAMPS=1
for form in ${FORMS:-reg lds}; do
    [ "$form" = none ] && continue
    echo "== form $form, $N jobs"
    if [ "$form" = lds ]; then PCSF_EIGEN_LDS=1 run; elif [ "$form" = quad ]; then PCSF_EIGEN_FORM=2 run; else PCSF_EIGEN_FORM=1 run; fi
done
for lib in "$@"; do
    echo "== $lib, $N jobs"
    PCSF_LIBRARY=$PWD/$lib run
done
