"""ctypes binding of the CPU oracle (oracle/pcsf_oracle.h).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "_build", "libpcsf_oracle.so")

PO_MAX_NODES = 512
MLE, FIXED, OMEGA = 0, 1, 2


class PoTree(C.Structure):
    _fields_ = [("n", C.c_int32), ("nl", C.c_int32),
                ("child1", C.c_int16 * PO_MAX_NODES), ("child2", C.c_int16 * PO_MAX_NODES),
                ("parent", C.c_int16 * PO_MAX_NODES), ("sibling", C.c_int16 * PO_MAX_NODES),
                ("branch_len", C.c_float * PO_MAX_NODES), ("branch_len_d", C.c_double * PO_MAX_NODES),
                ("label", (C.c_char * 64) * PO_MAX_NODES)]


class PoModel(C.Structure):
    _fields_ = [("tree", PoTree),
                ("S_c", C.c_double * 4096), ("pi_c", C.c_double * 64),
                ("S_nc", C.c_double * 4096), ("pi_nc", C.c_double * 64)]


class PoQdiag(C.Structure):
    _fields_ = [("Q", C.c_double * 4096), ("lam", C.c_double * 64), ("Sr", C.c_double * 4096),
                ("Sinv", C.c_double * 4096), ("prior", C.c_double * 64)]


class PoTraceEntry(C.Structure):
    _fields_ = [("kind", C.c_int32), ("failed", C.c_int32), ("x", C.c_double), ("tree_scale", C.c_double),
                ("q_settings", C.c_double * 12), ("lpr", C.c_double)]


def build():
    """make -C oracle: called by tests/conftest.py at session start and by __graft_entry__.build(), never from lib()."""
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):  # no build step inside a (possibly profiled, GPU-initialised) process
            raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (or make -C oracle)" % LIB)
        _lib = C.CDLL(LIB)
        dp, u8p = C.POINTER(C.c_double), C.POINTER(C.c_uint8)
        _lib.po_model_open.argtypes = [C.c_char_p, C.POINTER(PoModel)]
        _lib.po_tree_parse.argtypes = [C.c_char_p, C.POINTER(PoTree)]
        _lib.po_run_tracks.argtypes = [C.POINTER(PoModel), u8p, C.c_size_t, C.c_size_t, dp]
        _lib.po_run.argtypes = [C.POINTER(PoModel), C.c_int, u8p, C.c_size_t, C.c_size_t, C.c_int, C.c_uint32,
                                C.POINTER(C.c_float), C.POINTER(C.c_float)]
        _lib.po_baseline_run_tracks.argtypes = [C.POINTER(PoModel), u8p, C.POINTER(C.c_uint64), C.c_size_t,
                                                C.c_size_t, dp, C.c_int]
        _lib.po_q_from_ecm.argtypes = [dp, dp, dp]
        _lib.po_q_from_omega.argtypes = [dp, dp, dp]
        _lib.po_eig.argtypes = [dp, dp, C.POINTER(PoQdiag)]
        _lib.po_make_P.argtypes = [C.POINTER(PoQdiag), C.c_double, dp]
        _lib.po_make_all_P.argtypes = [C.POINTER(PoQdiag), C.POINTER(PoTree), C.c_double, dp]
        _lib.po_lpr_leaves.argtypes = [C.POINTER(PoQdiag), C.POINTER(PoTree), C.c_double, u8p, C.c_size_t, C.c_size_t,
                                       C.c_int, dp, dp, dp]
        _lib.po_uniform_real.restype = C.c_double
        _lib.po_uniform_real.argtypes = [C.c_void_p, C.c_double, C.c_double]
        _lib.po_mt_seed.argtypes = [C.c_void_p, C.c_uint32]
        _lib.po_mt_next.restype = C.c_uint32
        _lib.po_mt_next.argtypes = [C.c_void_p]
        _lib.po_lpr_rho.restype = C.c_double
        _lib.po_lpr_rho.argtypes = [C.c_double]
        _lib.po_lpr_kappa.restype = C.c_double
        _lib.po_lpr_kappa.argtypes = [C.c_double]
        _lib.po_last_eval_count.restype = C.c_long
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _u8p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


class OracleModel:
    """A model loaded from <prefix>_coding.ECM / _noncoding.ECM / .nh by the oracle's own readers."""

    def __init__(self, prefix):
        self.m = PoModel()
        rc = lib().po_model_open(prefix.encode(), C.byref(self.m))
        if rc:
            raise RuntimeError("po_model_open(%s) = %d" % (prefix, rc))
        t = self.m.tree
        self.n, self.nl = t.n, t.nl
        self.child1 = np.array(t.child1[:t.n], dtype=np.int16)
        self.child2 = np.array(t.child2[:t.n], dtype=np.int16)
        self.parent = np.array(t.parent[:t.n], dtype=np.int16)
        self.branch_len = np.array(t.branch_len[:t.n], dtype=np.float32)
        self.branch_len_d = np.array(t.branch_len_d[:t.n], dtype=np.float64)
        self.labels = [bytes(t.label[i]).split(b"\0")[0].decode() for i in range(t.n)]
        self.S_c = np.array(self.m.S_c, dtype=np.float64).reshape(64, 64)
        self.pi_c = np.array(self.m.pi_c, dtype=np.float64)
        self.S_nc = np.array(self.m.S_nc, dtype=np.float64).reshape(64, 64)
        self.pi_nc = np.array(self.m.pi_nc, dtype=np.float64)

    def run_tracks(self, codons):
        """codons: uint8 [nl, n_sites] -> float64 [n_sites] decibans (run.hpp:35-55)."""
        codons = np.ascontiguousarray(codons, dtype=np.uint8)
        assert codons.shape[0] == self.nl
        n = codons.shape[1]
        out = np.empty(n, dtype=np.float64)
        rc = lib().po_run_tracks(C.byref(self.m), _u8p(codons), n, max(n, 1), _dp(out))
        if rc:
            raise RuntimeError("po_run_tracks failed")
        return out

    def bls(self, nucleotides):
        """compute_bls_score<true> on ASCII rows uint8 [nl, n_cols] -> (block score, per-base float64[n_cols])."""
        nuc = np.ascontiguousarray(nucleotides, dtype=np.uint8)
        assert nuc.shape[0] == self.nl
        out = np.empty(nuc.shape[1], dtype=np.float64)
        fn = lib().po_bls
        fn.restype = C.c_double
        fn.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_double)]
        score = fn(C.addressof(self.m.tree), nuc.ctypes.data_as(C.c_char_p), nuc.strides[0], nuc.shape[1], _dp(out))
        if score == -1.0:
            raise ValueError("unexpected character in alignment")
        return score, out

    def run(self, algo, codons, compute_anc=True, seed=42):
        codons = np.ascontiguousarray(codons, dtype=np.uint8)
        n = codons.shape[1]
        ph, an = C.c_float(), C.c_float()
        lib().po_run(C.byref(self.m), algo, _u8p(codons), n, max(n, 1), int(compute_anc), seed, C.byref(ph), C.byref(an))
        return ph.value, an.value

    def run_trace(self, algo, codons, compute_anc=False, seed=42, cap=4096):
        """po_run plus every objective evaluation in order: (phylo, anc, [PoTraceEntry...])."""
        codons = np.ascontiguousarray(codons, dtype=np.uint8)
        n = codons.shape[1]
        buf = (PoTraceEntry * cap)()
        ph, an, cnt = C.c_float(), C.c_float(), C.c_int()
        fn = lib().po_run_trace
        fn.argtypes = [C.POINTER(PoModel), C.c_int, C.POINTER(C.c_uint8), C.c_size_t, C.c_size_t, C.c_int, C.c_uint32,
                       C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(PoTraceEntry), C.c_int, C.POINTER(C.c_int)]
        fn(C.byref(self.m), algo, _u8p(codons), n, max(n, 1), int(compute_anc), seed, C.byref(ph), C.byref(an), buf, cap, C.byref(cnt))
        assert cnt.value <= cap, "trace buffer too small"
        return ph.value, an.value, [buf[i] for i in range(cnt.value)]

    def run_blocks(self, algo, codons, site_offsets, compute_anc=True, seed=42, threads=0):
        """po_run_blocks: score-msa's block loop (every block alone, seeded alike), OpenMP over blocks."""
        codons = np.ascontiguousarray(codons, dtype=np.uint8)
        off = np.ascontiguousarray(site_offsets, dtype=np.uint64)
        nb = len(off) - 1
        ph = np.full(nb, np.nan, np.float32)
        an = np.full(nb, np.nan, np.float32)
        fn = lib().po_run_blocks
        fn.argtypes = [C.POINTER(PoModel), C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.c_size_t, C.c_size_t, C.c_int,
                       C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int]
        fn(C.byref(self.m), algo, _u8p(codons), off.ctypes.data_as(C.POINTER(C.c_uint64)), nb, codons.shape[1], int(compute_anc), seed,
           ph.ctypes.data_as(C.POINTER(C.c_float)), an.ctypes.data_as(C.POINTER(C.c_float)), threads)
        return ph, an

    def eig_of_settings(self, q_settings):
        """PoQdiag of the kappa/omega F3x4 rate matrix at q_settings[12] (po_q_from_omega + po_eig)."""
        qs = np.ascontiguousarray(q_settings, dtype=np.float64)
        Q, pi = np.empty((64, 64)), np.empty(64)
        lib().po_q_from_omega(_dp(qs), _dp(Q), _dp(pi))
        qd = PoQdiag()
        lib().po_eig(_dp(Q), _dp(pi), C.byref(qd))
        return qd

    def baseline_run_tracks(self, codons, site_offsets, threads=0):
        codons = np.ascontiguousarray(codons, dtype=np.uint8)
        off = np.ascontiguousarray(site_offsets, dtype=np.uint64)
        out = np.empty(codons.shape[1], dtype=np.float64)
        rc = lib().po_baseline_run_tracks(C.byref(self.m), _u8p(codons), off.ctypes.data_as(C.POINTER(C.c_uint64)),
                                          len(off) - 1, codons.shape[1], _dp(out), threads)
        if rc:
            raise RuntimeError("po_baseline_run_tracks failed")
        return out

    def qdiag(self, coding=True):
        Q = np.empty((64, 64))
        S, pi = (self.S_c, self.pi_c) if coding else (self.S_nc, self.pi_nc)
        S = np.ascontiguousarray(S)
        lib().po_q_from_ecm(_dp(S), _dp(pi), _dp(Q))
        qd = PoQdiag()
        lib().po_eig(_dp(Q), _dp(pi), C.byref(qd))
        return qd

    def all_P(self, qd, x=1.0):
        P = np.empty((self.n - 1, 64, 64))
        rc = lib().po_make_all_P(C.byref(qd), C.byref(self.m.tree), x, _dp(P))
        return rc, P
