cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_e -- python3 tools/bench_blocks.py - 0 1 nocpu > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
import numpy as np
f = glob.glob('/tmp/prof_e/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for name in ('eigen_omega_kernel', 'prune_factored', 'edge_scale'):
    d = np.array([(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in rows if name in r['Kernel_Name']])
    g = np.array([int(r['Grid_Size']) // 256 for r in rows if name in r['Kernel_Name']]) if 'Grid_Size' in rows[0] else None
    print(name, 'launches', len(d), 'total %.3f s' % (d.sum() / 1e3), 'ms quantiles 10/25/50/75/90/99/max:', np.round(np.quantile(d, [.1, .25, .5, .75, .9, .99, 1.0]), 3))
    if g is not None:
        print('   work-groups per launch quantiles 10/50/90/max:', np.quantile(g, [.1, .5, .9, 1.0]))
        for lo, hi in ((0, 8), (8, 64), (64, 256), (256, 10**9)):
            sel = (g >= lo) & (g < hi)
            if sel.any():
                print('   launches with %d..%d work-groups: %d, median %.3f ms, mean %.3f ms' % (lo, hi, sel.sum(), np.median(d[sel]), d[sel].mean()))
PY
