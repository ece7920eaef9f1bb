"""CPU-only: the product's lock-step optimiser (phylocsfpp_amd/csrc/optimizer.hpp, through the C-ABI)
and the oracle's fit_find_init + Brent (oracle/pcsf_oracle.c) are driven with the SAME analytic
objective and must take exactly the same evaluation points -- the decision logic is shared
bit for bit, so on the GPU the only difference left is rounding in the objective itself."""
import ctypes as C
import math

import numpy as np
import pytest

import oracle_binding as ob
import phylocsfpp_amd as pa

FN = C.CFUNCTYPE(C.c_double, C.c_double, C.c_void_p)


def _run(fn, init, lo, hi, seed=42, skip=0):
    out = []
    cb = FN(lambda x, _u: fn(x))
    for lib, name in ((pa.load_library(), "pcsf_debug_maxlik"), (ob.lib(), "po_debug_max_lik")):
        f = getattr(lib, name)
        f.argtypes = [FN, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_int, C.POINTER(C.c_double), C.c_int,
                      C.POINTER(C.c_int), C.POINTER(C.c_double)]
        xs = np.zeros(1200)
        n, last = C.c_int(), C.c_double()
        assert f(cb, None, init, lo, hi, seed, skip, xs.ctypes.data_as(C.POINTER(C.c_double)), len(xs), C.byref(n), C.byref(last)) == 0
        out.append((xs[:n.value].copy(), last.value))
    return out


OBJECTIVES = [
    ("interior quadratic in log x", lambda x: -(math.log(x) - 0.3) ** 2, 1.0, 1e-2, 10.0),
    ("likelihood-like", lambda x: -50.0 * x + 120.0 * math.log(x) - 3.0 / x, 1.0, 1e-2, 10.0),
    ("maximum at the upper end", lambda x: x, 1.0, 1e-2, 10.0),                      # restarts exhaust -> hi
    ("maximum at the lower end", lambda x: -x, 1.0, 1e-2, 10.0),
    ("flat (a block without codons)", lambda x: 0.0, 1.0, 1e-2, 10.0),
    ("narrow peak found by restarts", lambda x: -abs(math.log(x) - 1.5) ** 0.5, 1.0, 1e-3, 10.0),
    ("kappa-like range", lambda x: -(x - 2.8) ** 2 + math.log(x), 2.5, 1.0, 10.0),
    ("asymmetric", lambda x: -math.exp(0.7 * x) + 4 * x, 1.0, 1e-3, 10.0),
]


@pytest.mark.parametrize("name,fn,init,lo,hi", OBJECTIVES)
@pytest.mark.parametrize("skip", [0, 7])
def test_same_evaluation_points(name, fn, init, lo, hi, skip):
    (xs_p, last_p), (xs_o, last_o) = _run(fn, init, lo, hi, skip=skip)
    assert len(xs_p) == len(xs_o), name
    assert np.array_equal(xs_p, xs_o), name          # bit-identical points, hence identical decisions
    assert last_p == last_o


def test_result_is_last_evaluation_not_best():
    # the reference returns params.lpr of the LAST call (src/fixed_lik.hpp:542-543)
    fn = lambda x: -(math.log(x) - 0.3) ** 2
    (xs, last), _ = _run(fn, 1.0, 1e-2, 10.0)
    assert last == fn(xs[-1])
    assert 4 <= len(xs) <= 60


def test_restart_path_uses_libstdcxx_uniform_real():
    # 250 failed restarts + 3 bracket evaluations + the final re-evaluation = 254
    (xs, last), (xo, _) = _run(lambda x: x, 1.0, 1e-2, 10.0)
    assert len(xs) == 254 and xs[-1] == 10.0 and last == 10.0
    assert np.all((xs[3:253] > 1e-2) & (xs[3:253] < 10.0))
