"""Compares two .scores files row by row: python tools/c3_compare.py a.scores b.scores"""
import sys
import numpy as np
def load(p):
    rows = [l.rstrip("\n").split("\t") for l in open(p) if not l.startswith("#")][1:]
    return np.array([[float(x) for x in r[4:]] for r in rows])
a, b = load(sys.argv[1]), load(sys.argv[2])
assert a.shape == b.shape, (a.shape, b.shape)
for col in range(a.shape[1]):
    x, y = a[:, col], b[:, col]
    both = np.isfinite(x) & np.isfinite(y)
    d = np.abs(x[both] - y[both])
    rel = d / np.maximum(1e-6, np.abs(y[both]))
    print("column %d: %d rows, nan pattern equal %s, |d| max %.3g median %.3g, rel<=1e-6: %d of %d, |d|<=0.316: %d" % (
        col, len(x), bool(np.array_equal(np.isnan(x), np.isnan(y))), d.max() if len(d) else 0, np.median(d) if len(d) else 0,
        int((rel <= 1e-6).sum()), int(both.sum()), int((d <= 0.316).sum())))
