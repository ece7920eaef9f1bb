#!/bin/bash
# Where an OMEGA round goes: kernel timeline (rocprofv3 --kernel-trace; the copy trace crashed the profiler at exit here) of bench_blocks.py on the 516-block
# file; prints, for the OMEGA part of the run, the time covered by each kernel / copies and the time covered by nothing.
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; rm -rf /tmp/ort
rocprofv3 --kernel-trace --output-format csv -d /tmp/ort -- python3 tools/bench_blocks.py - 0 1 nocpu > /tmp/ort.log 2>&1
grep "GPU " /tmp/ort.log
python3 - <<'PY'
import csv, glob
ev = []
for f in glob.glob('/tmp/ort/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-40:]))
for f in glob.glob('/tmp/ort/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'copy ' + r.get('Direction', '')))
ev.sort()
# the OMEGA part: from the eigen_omega_kernel launch one eighth into the list of eigen launches on (skips the warm-up call)
eig = [e for e in ev if 'eigen_omega' in e[2]]
t0 = eig[len(eig) // 8][0]
ev = [e for e in ev if e[0] >= t0]
t1 = max(e[1] for e in ev)
span = t1 - t0
# union coverage and per-name exclusive-ish totals
cover, cur_end, last = 0, t0, t0
for s, e, _ in ev:
    if s > cur_end:
        cur_end = s
    if e > cur_end:
        cover += e - cur_end
        cur_end = e
tot = {}
for s, e, n in ev:
    tot[n] = tot.get(n, 0) + (e - s)
print("OMEGA part: %.1f ms; something running on the device %.1f ms (%.0f %%), nothing %.1f ms" % (span / 1e6, cover / 1e6, 100.0 * cover / span, (span - cover) / 1e6))
for n, v in sorted(tot.items(), key=lambda kv: -kv[1])[:8]:
    print("  %-44s %8.1f ms  (%d)" % (n, v / 1e6, sum(1 for e in ev if e[2] == n)))
PY
