"""Device timeline from a rocprofv3 --kernel-trace directory: python tools/kernel_timeline.py <dir> [name filter for the window]
Prints the span from the first to the last kernel (or of the kernels matching the filter), the time covered by at least one kernel,
per-kernel totals, and how the uncovered time splits into gaps by length."""
import csv, glob, sys
ev = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-44:], int(r.get('Grid_Size', 0) or 0), int(r.get('Workgroup_Size', 0) or 1)))
ev.sort()
flt = sys.argv[2] if len(sys.argv) > 2 else None
win = [e for e in ev if flt in e[2]] if flt else ev
t0, t1 = min(e[0] for e in win), max(e[1] for e in win)
ev = [e for e in ev if e[0] >= t0 and e[1] <= t1]
cover, cur_end, gaps = 0, t0, []
for s, e, *_ in ev:
    if s > cur_end:
        gaps.append(s - cur_end)
        cur_end = s
    if e > cur_end:
        cover += e - cur_end
        cur_end = e
span = t1 - t0
print("window %.1f ms; at least one kernel running %.1f ms (%.0f %%), nothing %.1f ms in %d gaps" % (span / 1e6, cover / 1e6, 100.0 * cover / span, (span - cover) / 1e6, len(gaps)))
for lo, hi in ((0, 5e3), (5e3, 2e4), (2e4, 1e5), (1e5, 1e6), (1e6, 1e12)):
    g = [x for x in gaps if lo <= x < hi]
    print("  gaps of %7.0f..%-9.0f us: %5d, %.1f ms" % (lo / 1e3, hi / 1e3, len(g), sum(g) / 1e6))
tot = {}
for s, e, n, grid, wg in ev:
    t = tot.setdefault(n, [0, 0, 0])
    t[0] += e - s
    t[1] += 1
    t[2] += grid // max(wg, 1)
for n, (v, c, wgs) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:8]:
    print("  %-46s %9.1f ms  %6d launches  %9d work-groups (%.0f per launch, %.1f us per work-group-slot of 512)" % (n, v / 1e6, c, wgs, wgs / c, v / 1e3 / max(1.0, wgs / 512.0)))
