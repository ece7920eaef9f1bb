"""phylocsfpp_amd -- MI355X (gfx950) PhyloCSF scoring core.

Thin ctypes binding of the C-ABI in include/phylocsf_amd.h (libphylocsf_amd.so, hand-written
HIP kernels).  Mirrors the two entry points of the reference's scoring core,
``run_tracks`` (src/run.hpp:35) and ``run`` (src/run.hpp:57).  There is no CPU implementation
in this package: without the built extension or without a gfx950 device every call raises.
"""
import ctypes as C
import os

import numpy as np

__all__ = ["ScoringContext", "Model", "load_model", "PcsfError", "library_path", "load_library", "FIXED", "MLE", "OMEGA"]

_HERE = os.path.dirname(os.path.abspath(__file__))
# PCSF_LIBRARY: load another build of the same HIP library (kernel experiments); never a CPU implementation
_LIB_PATH = os.environ.get("PCSF_LIBRARY") or os.path.join(_HERE, "libphylocsf_amd.so")

MLE, FIXED, OMEGA = 0, 1, 2  # algorithm_t, src/run.hpp:16-21


class PcsfError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("phylocsf_amd error %d: %s" % (code, msg))
        self.code = code


class _AlignmentDesc(C.Structure):
    _fields_ = [("col_offset", C.c_uint64), ("len", C.c_uint32), ("start_pos", C.c_uint64), ("chrom_len", C.c_uint64)]


class _ModelDesc(C.Structure):
    _fields_ = [("n_nodes", C.c_int32),
                ("child1", C.POINTER(C.c_int16)), ("child2", C.POINTER(C.c_int16)),
                ("branch_len", C.POINTER(C.c_float)),
                ("S_coding", C.POINTER(C.c_double)), ("pi_coding", C.POINTER(C.c_double)),
                ("S_noncoding", C.POINTER(C.c_double)), ("pi_noncoding", C.POINTER(C.c_double))]


_lib = None


def library_path():
    return _LIB_PATH


def load_library():
    """Loads the in-tree HIP extension; raises (never falls back) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise ImportError("%s is missing: build it with `python -m phylocsfpp_amd._build` "
                          "(hipcc, --offload-arch=gfx950); there is no CPU fallback" % _LIB_PATH)
    # PyTorch-ROCm ships its own HIP runtime; load it first so that this process has ONE runtime
    # (initialising the system copy first makes torch.cuda report "No HIP GPUs are available").
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(_LIB_PATH)
    vp, dp, u8p, fp = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint8), C.POINTER(C.c_float)
    lib.pcsf_version.restype = C.c_char_p
    lib.pcsf_create.argtypes = [C.POINTER(vp), C.POINTER(_ModelDesc), C.c_int]
    lib.pcsf_destroy.argtypes = [vp]
    lib.pcsf_last_error.restype = C.c_char_p
    lib.pcsf_last_error.argtypes = [vp]
    lib.pcsf_n_leaves.argtypes = [vp]
    lib.pcsf_score_sites_fixed.argtypes = [vp, u8p, C.c_size_t, C.c_size_t, dp]
    lib.pcsf_score_sites_fixed_dev.argtypes = [vp, vp, C.c_size_t, C.c_size_t, vp, vp, vp]
    lib.pcsf_score_blocks.argtypes = [vp, C.c_int, u8p, C.POINTER(C.c_uint64), C.c_size_t, C.c_size_t, C.c_int,
                                      C.c_uint32, fp, fp]
    lib.pcsf_prepare_block_scoring.argtypes = [vp, C.c_int]
    lib.pcsf_set_block_kernel.argtypes = [vp, C.c_int]
    lib.pcsf_debug_maxlik_device.argtypes = [vp, C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_int, dp, C.c_int, dp, C.c_int,
                                             C.POINTER(C.c_int), dp]
    lib.pcsf_choose_block_kernel.argtypes = [C.c_int32, u8p, C.POINTER(C.c_uint64), C.c_size_t, C.c_size_t]
    lib.pcsf_score_alignments_6frames.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(_AlignmentDesc), C.c_size_t, dp, C.c_size_t,
                                                  C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.pcsf_score_alignments_6frames_bls.argtypes = lib.pcsf_score_alignments_6frames.argtypes + [dp]
    lib.pcsf_score_alignments_6frames_wig.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(_AlignmentDesc), C.c_size_t, C.POINTER(C.c_char_p),
                                                      C.c_float, C.c_int, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_size_t)]
    lib.pcsf_wig_text_estimate.argtypes = [C.c_size_t, C.c_size_t]
    lib.pcsf_wig_text_estimate.restype = C.c_size_t
    lib.pcsf_fetch_wig_text.argtypes = [vp, C.POINTER(C.c_char_p)]
    lib.pcsf_debug_wig_format.argtypes = [vp, fp, C.c_size_t, C.c_int, C.c_char_p, C.POINTER(C.c_uint32)]
    lib.pcsf_comm_unique_id.argtypes = [C.c_char_p]
    lib.pcsf_comm_create.argtypes = [vp, C.c_int, C.c_int, C.c_char_p]
    lib.pcsf_comm_destroy.argtypes = [vp]
    lib.pcsf_wig_retain_last.argtypes = [vp, C.c_uint64]
    lib.pcsf_wig_retain_host.argtypes = [vp, C.c_uint64, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64)]
    lib.pcsf_wig_retained_bytes.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.pcsf_wig_retain_reset.argtypes = [vp]
    lib.pcsf_comm_gather_wig.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
    lib.pcsf_comm_fetch_gathered.argtypes = [vp, C.c_int, C.c_int, C.c_char_p]
    lib.pcsf_set_bls_branch_lengths.argtypes = [vp, dp]
    lib.pcsf_branch_length_scores.argtypes = [vp, C.c_char_p, C.c_size_t, C.c_size_t, dp]
    lib.pcsf_hmm_smooth_runs.argtypes = [vp, dp, dp, dp, C.POINTER(C.c_uint64), C.c_size_t, dp, u8p, dp, C.POINTER(C.c_uint32)]
    lib.pcsf_block_loglik.argtypes = [vp, C.c_int, dp, dp, u8p, C.POINTER(C.c_uint64), C.c_size_t, C.c_size_t, dp,
                                      C.POINTER(C.c_int32)]
    lib.pcsf_block_loglik_eig.argtypes = [vp, dp, dp, dp, dp, C.c_size_t, dp, u8p, C.POINTER(C.c_uint64), C.c_size_t, C.c_size_t, dp,
                                          C.POINTER(C.c_int32)]
    lib.pcsf_set_timing.argtypes = [vp, C.c_int]
    lib.pcsf_last_kernel_ms.restype = C.c_double
    lib.pcsf_last_kernel_ms.argtypes = [vp]
    lib.pcsf_get_branch_matrices.argtypes = [vp, dp]
    lib.pcsf_build_schedule.argtypes = [C.c_int32, C.POINTER(C.c_int16), C.POINTER(C.c_int16), C.POINTER(C.c_uint32),
                                        C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                        C.POINTER(C.c_int32)]
    lib.pcsf_model_open.argtypes = [C.POINTER(vp), C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p]
    lib.pcsf_model_close.argtypes = [vp]
    lib.pcsf_model_describe.argtypes = [vp, C.POINTER(_ModelDesc)]
    lib.pcsf_model_branch_len_double.restype = dp
    lib.pcsf_model_branch_len_double.argtypes = [vp]
    lib.pcsf_model_parents.restype = C.POINTER(C.c_int16)
    lib.pcsf_model_parents.argtypes = [vp]
    lib.pcsf_model_label.restype = C.c_char_p
    lib.pcsf_model_label.argtypes = [vp, C.c_int32]
    lib.pcsf_model_leaf_of.restype = C.c_int32
    lib.pcsf_model_leaf_of.argtypes = [vp, C.c_char_p]
    _lib = lib
    return lib


class Model:
    """What load_model hands to the scoring core (reference src/models.hpp:1742-1857): the flattened tree in
    newick_flatten numbering, both ECMs, the species labels.  Arrays are numpy copies; `leaf_of(name)` resolves a
    species name of an alignment (tree labels and their other names) to a leaf index."""

    def __init__(self, prefix, aliases=None, species=None, mapping=None):
        lib = load_library()
        enc = lambda x: None if x is None else os.fsencode(x)
        h = C.c_void_p()
        rc = lib.pcsf_model_open(C.byref(h), os.fsencode(prefix), enc(aliases), enc(species), enc(mapping))
        if rc:
            raise PcsfError(rc, lib.pcsf_last_error(None).decode())
        try:
            d = _ModelDesc()
            lib.pcsf_model_describe(h, C.byref(d))
            n = self.n = self.n_nodes = int(d.n_nodes)
            self.nl = self.n_leaves = (n + 1) // 2
            arr = lambda p, k, t: np.ctypeslib.as_array(p, shape=(k,)).astype(t, copy=True)
            self.child1 = arr(d.child1, n, np.int16)
            self.child2 = arr(d.child2, n, np.int16)
            self.branch_len = arr(d.branch_len, n, np.float32)
            self.branch_len_d = arr(lib.pcsf_model_branch_len_double(h), n, np.float64)
            self.parent = arr(lib.pcsf_model_parents(h), n, np.int16)
            self.S_c = arr(d.S_coding, 4096, np.float64).reshape(64, 64)
            self.pi_c = arr(d.pi_coding, 64, np.float64)
            self.S_nc = arr(d.S_noncoding, 4096, np.float64).reshape(64, 64)
            self.pi_nc = arr(d.pi_noncoding, 64, np.float64)
            self.labels = [lib.pcsf_model_label(h, i).decode() for i in range(n)]
        except Exception:
            lib.pcsf_model_close(h)
            raise
        self._h, self._lib = h, lib
        self.prefix = prefix

    def leaf_of(self, species_name):
        return int(self._lib.pcsf_model_leaf_of(self._h, species_name.encode()))

    def context(self, device=0, with_bls=False):
        """A ScoringContext of this model on GPU `device` (pcsf_create)."""
        ctx = ScoringContext(self.child1, self.child2, self.branch_len, self.S_c, self.pi_c, self.S_nc, self.pi_nc, device=device)
        if with_bls:
            ctx.set_bls_branch_lengths(self.branch_len_d)
        return ctx

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pcsf_model_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def load_model(prefix, aliases=None, species=None, mapping=None):
    """<prefix>_coding.ECM, <prefix>_noncoding.ECM, <prefix>.nh -> Model (C-ABI pcsf_model_open; host-only)."""
    return Model(prefix, aliases=aliases, species=species, mapping=mapping)


def build_schedule(child1, child2):
    """Host-only: the pruning program the kernel interprets for this tree (no GPU needed).
    Returns (program uint32[n_ops], slot_node int32[n_slots], stack_depth)."""
    lib = load_library()
    c1 = np.ascontiguousarray(child1, dtype=np.int16)
    c2 = np.ascontiguousarray(child2, dtype=np.int16)
    n = len(c1)
    prog = np.zeros(4 * n + 8, dtype=np.uint32)
    slots = np.zeros(n, dtype=np.int32)
    n_ops, n_slots, depth = C.c_int32(), C.c_int32(), C.c_int32()
    rc = lib.pcsf_build_schedule(n, c1.ctypes.data_as(C.POINTER(C.c_int16)), c2.ctypes.data_as(C.POINTER(C.c_int16)),
                                 prog.ctypes.data_as(C.POINTER(C.c_uint32)), len(prog),
                                 slots.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(n_ops), C.byref(n_slots),
                                 C.byref(depth))
    if rc:
        raise PcsfError(rc, lib.pcsf_last_error(None).decode())
    return prog[:n_ops.value].copy(), slots[:n_slots.value].copy(), depth.value


def build_cherry_schedule(child1, child2, triples=False):
    """Host-only: the program tracks mode runs when the tree has cherries (every cherry below the root a leaf with a pair table; with
    triples=True also every node over a cherry and a leaf, a 65^3 table).
    Returns (program, slot_node -- nodes of the ORIGINAL tree --, stack_depth, pairs int32[n_pairs, 3] = leaf A, leaf B, cherry node[,
    triples int32[n_triples, 3] = pair index, third leaf, node]), or None when the tree has no such cherry.  An operand >= 8192 in an op
    word names pair operand - 8192, >= 12288 triple operand - 12288."""
    lib = load_library()
    c1 = np.ascontiguousarray(child1, dtype=np.int16)
    c2 = np.ascontiguousarray(child2, dtype=np.int16)
    n = len(c1)
    prog = np.zeros(4 * n + 8, dtype=np.uint32)
    slots = np.zeros(n, dtype=np.int32)
    pairs = np.zeros(3 * ((n + 1) // 2), dtype=np.int32)
    trips = np.zeros(3 * ((n + 1) // 2), dtype=np.int32)
    n_ops, n_slots, depth, n_pairs, n_trips = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    i16, i32 = C.POINTER(C.c_int16), C.POINTER(C.c_int32)
    lib.pcsf_debug_cherry_schedule.argtypes = [C.c_int32, i16, i16, C.POINTER(C.c_uint32), C.c_int32, i32, i32, i32, i32, i32, i32, i32, i32]
    rc = lib.pcsf_debug_cherry_schedule(n, c1.ctypes.data_as(i16), c2.ctypes.data_as(i16), prog.ctypes.data_as(C.POINTER(C.c_uint32)), len(prog),
                                        slots.ctypes.data_as(i32), C.byref(n_ops), C.byref(n_slots), C.byref(depth), pairs.ctypes.data_as(i32), C.byref(n_pairs),
                                        trips.ctypes.data_as(i32) if triples else None, C.byref(n_trips) if triples else None)
    if rc:
        raise PcsfError(rc, lib.pcsf_last_error(None).decode())
    if n_pairs.value == 0:
        return None
    out = (prog[:n_ops.value].copy(), slots[:n_slots.value].copy(), depth.value, pairs[:3 * n_pairs.value].reshape(-1, 3).copy())
    return out + (trips[:3 * n_trips.value].reshape(-1, 3).copy(),) if triples else out


def build_table_schedule(child1, child2, max_leaves=3):
    """Host-only: the program tracks mode runs with subtree tables of up to max_leaves (2, 3 or 4) leaves.  Returns (program, slot_node -- nodes
    of the ORIGINAL tree --, stack_depth, {operand id: node whose message to its parent the operand's table holds}), or None when the tree
    has no cherry below the root."""
    lib = load_library()
    c1 = np.ascontiguousarray(child1, dtype=np.int16)
    c2 = np.ascontiguousarray(child2, dtype=np.int16)
    n = len(c1)
    prog = np.zeros(4 * n + 8, dtype=np.uint32)
    slots = np.zeros(n, dtype=np.int32)
    ops = np.zeros(2 * (n + 1), dtype=np.int32)
    n_ops, n_slots, depth, n_operands = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    i16, i32 = C.POINTER(C.c_int16), C.POINTER(C.c_int32)
    lib.pcsf_debug_table_schedule.argtypes = [C.c_int32, i16, i16, C.c_int32, C.POINTER(C.c_uint32), C.c_int32, i32, i32, i32, i32, i32, i32]
    rc = lib.pcsf_debug_table_schedule(n, c1.ctypes.data_as(i16), c2.ctypes.data_as(i16), int(max_leaves), prog.ctypes.data_as(C.POINTER(C.c_uint32)), len(prog),
                                       slots.ctypes.data_as(i32), C.byref(n_ops), C.byref(n_slots), C.byref(depth), ops.ctypes.data_as(i32), C.byref(n_operands))
    if rc:
        raise PcsfError(rc, lib.pcsf_last_error(None).decode())
    if n_operands.value == 0:
        return None
    return (prog[:n_ops.value].copy(), slots[:n_slots.value].copy(), depth.value,
            {int(ops[2 * k]): int(ops[2 * k + 1]) for k in range(n_operands.value)})


class ScoringContext:
    """One model resident on one GPU (replaces the per-call PhyloCSFModel_make of src/run.hpp:38-39).

    child1/child2/branch_len: the flattened tree in newick_flatten order (src/newick.hpp:170-229);
    S_*/pi_*: empirical_codon_model::matrix / ::codon_freq (src/ecm.hpp:15-19)."""

    def __init__(self, child1, child2, branch_len, S_coding, pi_coding, S_noncoding, pi_noncoding, device=0):
        lib = load_library()
        self._lib = lib
        self._keep = dict(
            c1=np.ascontiguousarray(child1, dtype=np.int16), c2=np.ascontiguousarray(child2, dtype=np.int16),
            bl=np.ascontiguousarray(branch_len, dtype=np.float32),
            sc=np.ascontiguousarray(S_coding, dtype=np.float64).reshape(-1),
            pc=np.ascontiguousarray(pi_coding, dtype=np.float64),
            sn=np.ascontiguousarray(S_noncoding, dtype=np.float64).reshape(-1),
            pn=np.ascontiguousarray(pi_noncoding, dtype=np.float64))
        k = self._keep
        n = len(k["c1"])
        if not (len(k["c2"]) == n == len(k["bl"])) or k["sc"].size != 4096 or k["sn"].size != 4096 \
                or k["pc"].size != 64 or k["pn"].size != 64:
            raise ValueError("inconsistent model array sizes")
        desc = _ModelDesc(n, k["c1"].ctypes.data_as(C.POINTER(C.c_int16)), k["c2"].ctypes.data_as(C.POINTER(C.c_int16)),
                          k["bl"].ctypes.data_as(C.POINTER(C.c_float)),
                          k["sc"].ctypes.data_as(C.POINTER(C.c_double)), k["pc"].ctypes.data_as(C.POINTER(C.c_double)),
                          k["sn"].ctypes.data_as(C.POINTER(C.c_double)), k["pn"].ctypes.data_as(C.POINTER(C.c_double)))
        h = C.c_void_p()
        rc = lib.pcsf_create(C.byref(h), C.byref(desc), int(device))
        if rc:
            raise PcsfError(rc, lib.pcsf_last_error(None).decode())
        self._h = h
        self.device = int(device)
        self.n_nodes = n
        self.n_leaves = (n + 1) // 2

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pcsf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise PcsfError(rc, self._lib.pcsf_last_error(self._h).decode())

    # ---- run_tracks ----------------------------------------------------------------------
    def tracks_program_info(self):
        """what the tracks-mode kernel of this context runs: products per site and model of the full program and of the program it runs, cherries
        with a pair table (0: the full program), rows per table set"""
        out = (C.c_int64 * 6)()
        self._lib.pcsf_tracks_program_info.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        self._check(self._lib.pcsf_tracks_program_info(self._h, out))
        return {"products_full": int(out[0]), "products_run": int(out[1]), "cherries": int(out[2]), "rows_per_table_set": int(out[3]), "triples": int(out[4]),
                "quads": int(out[5])}

    def score_sites_fixed(self, codons):
        """codons: uint8 ndarray [n_leaves, n_sites] (host) -> float64 ndarray [n_sites] of
        10*(ln z_c - ln z_nc)/ln 10, the vector run_tracks returns (src/run.hpp:51-54)."""
        codons = np.asarray(codons)
        if codons.dtype != np.uint8 or codons.ndim != 2 or codons.shape[0] != self.n_leaves:
            raise ValueError("codons must be uint8 [n_leaves=%d, n_sites]" % self.n_leaves)
        n = codons.shape[1]
        out = np.empty(n, dtype=np.float64)
        if n == 0:
            return out
        if codons.strides[1] != 1:
            codons = np.ascontiguousarray(codons)
        self._check(self._lib.pcsf_score_sites_fixed(self._h, codons.ctypes.data_as(C.POINTER(C.c_uint8)), n,
                                                     codons.strides[0], out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def score_sites_fixed_device(self, d_codons, n_sites, leaf_stride, d_out, d_lnz=0, stream=0):
        """Raw device-pointer form (ints): enqueues on `stream` (hipStream_t as int) without synchronising."""
        self._check(self._lib.pcsf_score_sites_fixed_dev(self._h, C.c_void_p(d_codons), n_sites, leaf_stride,
                                                         C.c_void_p(d_out), C.c_void_p(d_lnz or None),
                                                         C.c_void_p(stream or None)))

    def score_sites_fixed_torch(self, codons, out=None, lnz=None):
        """codons: torch.uint8 CUDA tensor [n_leaves, n_sites] on this context's GPU.  Launches on
        torch's current stream; returns a float64 CUDA tensor [n_sites]."""
        import torch
        if codons.dtype != torch.uint8 or codons.dim() != 2 or codons.shape[0] != self.n_leaves or not codons.is_cuda:
            raise ValueError("codons must be a CUDA uint8 tensor [n_leaves, n_sites]")
        if codons.stride(1) != 1:
            codons = codons.contiguous()
        n = codons.shape[1]
        if out is None:
            out = torch.empty(n, dtype=torch.float64, device=codons.device)
        stream = torch.cuda.current_stream(codons.device).cuda_stream
        self.score_sites_fixed_device(codons.data_ptr(), n, codons.stride(0), out.data_ptr(),
                                      lnz.data_ptr() if lnz is not None else 0, stream)
        return out

    # ---- branch-length score ----------------------------------------------------------------
    def set_bls_branch_lengths(self, branch_len):
        """Double-precision branch lengths [n_nodes] for the branch-length score (the reference's newick_node)."""
        bl = np.ascontiguousarray(branch_len, dtype=np.float64)
        self._check(self._lib.pcsf_set_bls_branch_lengths(self._h, bl.ctypes.data_as(C.POINTER(C.c_double))))

    def branch_length_scores(self, nucleotides):
        """compute_bls_score per column of the ASCII rows nucleotides[n_leaves, n_cols] -> float64[n_cols]."""
        nuc = np.ascontiguousarray(nucleotides, dtype=np.uint8)
        assert nuc.ndim == 2 and nuc.shape[0] == self.n_leaves
        out = np.empty(nuc.shape[1], dtype=np.float64)
        self._check(self._lib.pcsf_branch_length_scores(self._h, nuc.ctypes.data_as(C.c_char_p), nuc.strides[0], nuc.shape[1],
                                                        out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def hmm_smooth_runs(self, init, trans, emissions, run_offsets, viterbi=True):
        """The smoothing pass's recurrences for a batch of runs (pcsf_hmm_smooth_runs): coding posterior float64[n] and, with
        viterbi, the packed back pointers uint8[n] and the last rescaled Viterbi vector of every run float64[n_runs, 4];
        the fourth value is the call's statistics (chunks, forward reruns / passes, backward ..., viterbi ...)."""
        dp = C.POINTER(C.c_double)
        init = np.ascontiguousarray(init, dtype=np.float64)
        trans = np.ascontiguousarray(trans, dtype=np.float64)
        em = np.ascontiguousarray(emissions, dtype=np.float64)
        offs = np.ascontiguousarray(run_offsets, dtype=np.uint64)
        assert init.shape == (4,) and trans.shape == (4, 4) and offs.ndim == 1 and len(offs) >= 1 and int(offs[-1]) == len(em)
        n_runs = len(offs) - 1
        post = np.empty(len(em), dtype=np.float64)
        back = np.zeros(len(em), dtype=np.uint8) if viterbi else None
        last = np.zeros((n_runs, 4), dtype=np.float64) if viterbi else None
        stats = np.zeros(7, dtype=np.uint32)
        self._check(self._lib.pcsf_hmm_smooth_runs(self._h, init.ctypes.data_as(dp), trans.ctypes.data_as(dp), em.ctypes.data_as(dp),
                                                   offs.ctypes.data_as(C.POINTER(C.c_uint64)), n_runs, post.ctypes.data_as(dp),
                                                   back.ctypes.data_as(C.POINTER(C.c_uint8)) if viterbi else None,
                                                   last.ctypes.data_as(dp) if viterbi else None, stats.ctypes.data_as(C.POINTER(C.c_uint32))))
        return post, back, last, stats

    def score_alignments_6frames(self, nucleotides, alignments, with_bls=False):
        """Six-frame front end (device translation + run_tracks).  nucleotides: uint8/bytes ndarray
        [n_leaves, n_cols] of ASCII rows (reference gaps removed); alignments: list of
        (col_offset, len, start_pos, chrom_len).  Returns a list (per alignment) of 6 float64 arrays in
        forward genome order (+1,+2,+3,-1,-2,-3) and the skip_bases array [n, 6]; with_bls adds the per-column
        branch-length scores float64[n_cols] as a third value."""
        nuc = np.ascontiguousarray(nucleotides, dtype=np.uint8)
        assert nuc.ndim == 2 and nuc.shape[0] == self.n_leaves
        n = len(alignments)
        desc = (_AlignmentDesc * n)(*[_AlignmentDesc(int(a), int(b), int(c), int(d)) for a, b, c, d in alignments])
        cap = int(sum(2 * int(a[1]) for a in alignments)) + 1
        out = np.empty(cap, dtype=np.float64)
        offs = np.zeros(n * 6, dtype=np.uint64)
        ncod = np.zeros(n * 6, dtype=np.uint32)
        skip = np.zeros(n * 6, dtype=np.uint32)
        bls = np.zeros(nuc.shape[1], dtype=np.float64) if with_bls else None
        self._check(self._lib.pcsf_score_alignments_6frames_bls(
            self._h, nuc.ctypes.data_as(C.c_char_p), nuc.strides[0], desc, n, out.ctypes.data_as(C.POINTER(C.c_double)), cap,
            offs.ctypes.data_as(C.POINTER(C.c_uint64)), ncod.ctypes.data_as(C.POINTER(C.c_uint32)),
            skip.ctypes.data_as(C.POINTER(C.c_uint32)), bls.ctypes.data_as(C.POINTER(C.c_double)) if with_bls else None))
        res = [[out[int(offs[k * 6 + f]):int(offs[k * 6 + f]) + int(ncod[k * 6 + f])].copy() for f in range(6)] for k in range(n)]
        if with_bls:
            return res, skip.reshape(n, 6), bls
        return res, skip.reshape(n, 6)

    # ---- build-tracks as one process per GPU: the text kept on the device, gathered over RCCL (csrc/comm.cpp) ----
    @staticmethod
    def comm_unique_id():
        """rank 0: the communicator's 128-byte identifier (ncclGetUniqueId), to be handed to the other ranks out of band"""
        lib = load_library()
        buf = C.create_string_buffer(128)
        rc = lib.pcsf_comm_unique_id(buf)
        if rc != 0:
            raise PcsfError(rc, "pcsf_comm_unique_id failed (librccl.so not found?)")
        return buf.raw

    def comm_create(self, rank, world, comm_id):
        self._world = int(world)
        self._check(self._lib.pcsf_comm_create(self._h, int(rank), int(world), bytes(comm_id)))

    def comm_destroy(self):
        self._check(self._lib.pcsf_comm_destroy(self._h))

    def wig_retain_last(self, order):
        """keeps the text of the last score_alignments_6frames_wig call on the device under the key `order`"""
        self._check(self._lib.pcsf_wig_retain_last(self._h, int(order)))

    def wig_retain_host(self, order, texts):
        """adds host-formatted text (7 bytes objects) to the kept stream under the key `order`"""
        assert len(texts) == 7
        ptr = (C.c_char_p * 7)(*[bytes(t) if len(t) else None for t in texts])
        n = (C.c_uint64 * 7)(*[len(t) for t in texts])
        self._check(self._lib.pcsf_wig_retain_host(self._h, int(order), ptr, n))

    def wig_retained_bytes(self):
        n = (C.c_uint64 * 7)()
        self._check(self._lib.pcsf_wig_retained_bytes(self._h, n))
        return [int(v) for v in n]

    def comm_gather_wig(self, root=0, include_root=False):
        """collective: every sender's kept text to the root's GPU; returns bytes[world][7]"""
        n = (C.c_uint64 * (7 * self._world))()
        self._check(self._lib.pcsf_comm_gather_wig(self._h, int(root), int(bool(include_root)), n))
        return [[int(n[r * 7 + f]) for f in range(7)] for r in range(self._world)]

    def comm_fetch_gathered(self, rank, file, n_bytes):
        buf = C.create_string_buffer(max(int(n_bytes), 1))
        self._check(self._lib.pcsf_comm_fetch_gathered(self._h, int(rank), int(file), buf))
        return buf.raw[:int(n_bytes)]

    def score_alignments_6frames_wig(self, nucleotides, alignments, chroms, power_threshold, want_power=True, want_raw=True, text_capacity=None,
                                     retain_order=None):
        """The same call for a caller that only writes: the text of build-tracks' seven wig files for this batch (formatted on the
        device, csrc/wig.hip).  chroms: one chromosome name per alignment.  Returns (list of 7 bytes objects -- power track, then
        +1 +2 +3 -1 -2 -3 --, number of codon-frames scored); None when the device left the batch to the host formatter.  text_capacity: bytes
        of the buffer handed to the call (default: pcsf_wig_text_estimate); a text that does not fit is fetched afterwards."""
        nuc = np.ascontiguousarray(nucleotides, dtype=np.uint8)
        assert nuc.ndim == 2 and nuc.shape[0] == self.n_leaves
        n = len(alignments)
        desc = (_AlignmentDesc * n)(*[_AlignmentDesc(int(a), int(b), int(c), int(d)) for a, b, c, d in alignments])
        names = (C.c_char_p * n)(*[c.encode() if isinstance(c, str) else bytes(c) for c in chroms])
        sizes = (C.c_uint64 * 7)()
        frames = C.c_size_t(0)
        cap = int(text_capacity) if text_capacity is not None else int(self._lib.pcsf_wig_text_estimate(nuc.shape[1], n))
        text = C.create_string_buffer(max(cap, 1))
        rc = self._lib.pcsf_score_alignments_6frames_wig(self._h, nuc.ctypes.data_as(C.c_char_p), nuc.strides[0], desc, n, names,
                                                         float(power_threshold), int(want_power), int(want_raw), text if cap else None, cap,
                                                         sizes, C.byref(frames))
        if rc == -5:   # PCSF_ERR_UNSUPPORTED
            return None
        self._check(rc)
        if retain_order is not None:   # the call's text also stays on the device for the ranks' gather (wig_retain_last)
            self.wig_retain_last(retain_order)
        total = sum(int(s) for s in sizes)
        if total <= cap:       # the text came with the call
            raw, at, out = text.raw, 0, []
            for s in sizes:
                out.append(raw[at:at + int(s)])
                at += int(s)
            return out, int(frames.value)
        bufs = [C.create_string_buffer(max(int(s), 1)) for s in sizes]
        dst = (C.c_char_p * 7)(*[C.cast(b, C.c_char_p) for b in bufs])
        self._check(self._lib.pcsf_fetch_wig_text(self._h, dst))
        return [bufs[f].raw[:int(sizes[f])] for f in range(7)], int(frames.value)

    def debug_wig_format(self, values, decimals):
        """csrc/wig.hip's number formatter on float32 values: list of bytes (None: a value the device leaves to the host)."""
        v = np.ascontiguousarray(values, dtype=np.float32)
        text = C.create_string_buffer(max(32 * len(v), 1))
        lens = np.zeros(len(v), dtype=np.uint32)
        self._check(self._lib.pcsf_debug_wig_format(self._h, v.ctypes.data_as(C.POINTER(C.c_float)), len(v), int(decimals), text,
                                                    lens.ctypes.data_as(C.POINTER(C.c_uint32))))
        raw = text.raw
        return [None if int(l) == 0xffffffff else raw[32 * i:32 * i + int(l)] for i, l in enumerate(lens)]

    # ---- run -----------------------------------------------------------------------------
    def prepare_block_scoring(self, algo):
        """Optional: score_blocks(algo, ...) is coming -- its device queues are created in the background meanwhile."""
        self._check(self._lib.pcsf_prepare_block_scoring(self._h, int(algo)))

    BLOCK_KERNEL_AUTO, BLOCK_KERNEL_TILES, BLOCK_KERNEL_FACTORED = 0, 1, 2

    def set_block_kernel(self, kernel):
        """Pins the pruning kernel of MLE / OMEGA calls (pcsf_set_block_kernel).  AUTO lets every score_blocks call pick from its own
        list: a caller that cuts one list into several calls and wants identical scores pins choose_block_kernel(whole list)."""
        self._check(self._lib.pcsf_set_block_kernel(self._h, int(kernel)))

    def choose_block_kernel(self, codons, site_offsets):
        """The AUTO rule evaluated on a list (host only): BLOCK_KERNEL_TILES or BLOCK_KERNEL_FACTORED."""
        codons = np.ascontiguousarray(codons, dtype=np.uint8)
        off = np.ascontiguousarray(site_offsets, dtype=np.uint64)
        k = self._lib.pcsf_choose_block_kernel(self.n_leaves, codons.ctypes.data_as(C.POINTER(C.c_uint8)), off.ctypes.data_as(C.POINTER(C.c_uint64)),
                                               len(off) - 1, codons.strides[0])
        if k < 0:
            raise ValueError("pcsf_choose_block_kernel: bad argument")
        return k

    def debug_maxlik_device(self, init, lo, hi, values, seed=42, skip_draws=0):
        """Tests: the 1-D maximiser of the MLE / OMEGA engine on the device, fed with recorded objective values.  Returns
        (points asked for, number of requests or -1, last value)."""
        values = np.ascontiguousarray(values, dtype=np.float64)
        xs = np.zeros(len(values) + 8)
        n, last = C.c_int(), C.c_double()
        self._check(self._lib.pcsf_debug_maxlik_device(self._h, init, lo, hi, seed, skip_draws, values.ctypes.data_as(C.POINTER(C.c_double)), len(values),
                                                       xs.ctypes.data_as(C.POINTER(C.c_double)), len(xs), C.byref(n), C.byref(last)))
        return xs, n.value, last.value

    def score_blocks(self, algo, codons, site_offsets, compute_anc=True, seed=42):
        codons = np.ascontiguousarray(codons, dtype=np.uint8)
        off = np.ascontiguousarray(site_offsets, dtype=np.uint64)
        nb = len(off) - 1
        phylo = np.full(nb, np.nan, dtype=np.float32)
        anc = np.full(nb, np.nan, dtype=np.float32)
        self._check(self._lib.pcsf_score_blocks(self._h, int(algo), codons.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                off.ctypes.data_as(C.POINTER(C.c_uint64)), nb, codons.strides[0],
                                                int(compute_anc), seed, phylo.ctypes.data_as(C.POINTER(C.c_float)),
                                                anc.ctypes.data_as(C.POINTER(C.c_float))))
        return phylo, anc

    def block_loglik(self, model, codons, site_offsets, tree_scale, q_settings=None):
        """One objective evaluation per block (no prior term): model 0/1 = coding/non-coding ECM,
        2 = kappa/omega model with q_settings [n_blocks, 12].  Returns (lpr float64[n_blocks], failed)."""
        codons = np.ascontiguousarray(codons, dtype=np.uint8)
        off = np.ascontiguousarray(site_offsets, dtype=np.uint64)
        nb = len(off) - 1
        ts = np.ascontiguousarray(np.broadcast_to(np.asarray(tree_scale, dtype=np.float64), (nb,)))
        qs = None
        if q_settings is not None:
            qs = np.ascontiguousarray(np.broadcast_to(np.asarray(q_settings, dtype=np.float64), (nb, 12)))
        out = np.empty(nb, dtype=np.float64)
        failed = np.zeros(nb, dtype=np.int32)
        self._check(self._lib.pcsf_block_loglik(
            self._h, int(model), qs.ctypes.data_as(C.POINTER(C.c_double)) if qs is not None else None,
            ts.ctypes.data_as(C.POINTER(C.c_double)), codons.ctypes.data_as(C.POINTER(C.c_uint8)),
            off.ctypes.data_as(C.POINTER(C.c_uint64)), nb, codons.strides[0], out.ctypes.data_as(C.POINTER(C.c_double)),
            failed.ctypes.data_as(C.POINTER(C.c_int32))))
        return out, failed

    def block_loglik_eig(self, codons, site_offsets, tree_scale, right, left, lam, prior):
        """block_loglik with caller-supplied eigen-systems: right/left [k,64,64], lam/prior [k,64], k = 1 or n_blocks."""
        codons = np.ascontiguousarray(codons, dtype=np.uint8)
        off = np.ascontiguousarray(site_offsets, dtype=np.uint64)
        nb = len(off) - 1
        ts = np.ascontiguousarray(np.broadcast_to(np.asarray(tree_scale, dtype=np.float64), (nb,)))
        r = np.ascontiguousarray(np.asarray(right, dtype=np.float64).reshape(-1, 64, 64))
        l = np.ascontiguousarray(np.asarray(left, dtype=np.float64).reshape(-1, 64, 64))
        la = np.ascontiguousarray(np.asarray(lam, dtype=np.float64).reshape(-1, 64))
        pr = np.ascontiguousarray(np.asarray(prior, dtype=np.float64).reshape(-1, 64))
        k = r.shape[0]
        assert l.shape[0] == k and la.shape[0] == k and pr.shape[0] == k
        out = np.empty(nb, dtype=np.float64)
        failed = np.zeros(nb, dtype=np.int32)
        dp = C.POINTER(C.c_double)
        self._check(self._lib.pcsf_block_loglik_eig(
            self._h, r.ctypes.data_as(dp), l.ctypes.data_as(dp), la.ctypes.data_as(dp), pr.ctypes.data_as(dp), k,
            ts.ctypes.data_as(dp), codons.ctypes.data_as(C.POINTER(C.c_uint8)), off.ctypes.data_as(C.POINTER(C.c_uint64)), nb,
            codons.strides[0], out.ctypes.data_as(dp), failed.ctypes.data_as(C.POINTER(C.c_int32))))
        return out, failed

    # ---- diagnostics -----------------------------------------------------------------------
    def set_timing(self, enable=True):
        self._check(self._lib.pcsf_set_timing(self._h, int(enable)))

    def last_kernel_ms(self):
        return self._lib.pcsf_last_kernel_ms(self._h)

    def debug_eigen_omega(self, q_settings, warm_settings=None):
        """Eigen-systems of kappa/omega rate matrices from the device solver: q_settings [n, 12] ->
        (right [n,64,64], left [n,64,64], lambda [n,64], prior [n,64], failed [n])."""
        qs = np.ascontiguousarray(np.asarray(q_settings, dtype=np.float64).reshape(-1, 12))
        n = qs.shape[0]
        right, left = np.empty((n, 64, 64)), np.empty((n, 64, 64))
        lam, prior = np.empty((n, 64)), np.empty((n, 64))
        failed = np.zeros(n, dtype=np.int32)
        dp = C.POINTER(C.c_double)
        ws = None if warm_settings is None else np.ascontiguousarray(np.asarray(warm_settings, dtype=np.float64).reshape(-1, 12))
        self._lib.pcsf_debug_eigen_omega.argtypes = [C.c_void_p, dp, dp, C.c_size_t, dp, dp, dp, dp, C.POINTER(C.c_int32)]
        self._check(self._lib.pcsf_debug_eigen_omega(self._h, qs.ctypes.data_as(dp), None if ws is None else ws.ctypes.data_as(dp), n,
                                                     right.ctypes.data_as(dp), left.ctypes.data_as(dp),
                                                     lam.ctypes.data_as(dp), prior.ctypes.data_as(dp), failed.ctypes.data_as(C.POINTER(C.c_int32))))
        return right, left, lam, prior, failed

    def branch_matrices(self):
        out = np.empty((2, self.n_nodes - 1, 64, 64), dtype=np.float64)
        self._check(self._lib.pcsf_get_branch_matrices(self._h, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out
