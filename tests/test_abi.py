"""CPU: the C-ABI library loads and exports every entry point include/phylocsf_amd.h declares (no compute calls,
no GPU needed).  The header is the drop-in boundary."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "phylocsf_amd.h")


def _declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)          # comments mention function names too
    text = re.sub(r"//[^\n]*", " ", text)
    names = sorted(set(re.findall(r"\b(pcsf_[a-z0-9_]+)\s*\(", text)))
    return names


def test_header_declares_the_expected_entry_points():
    names = _declared_functions()
    for must in ("pcsf_create", "pcsf_destroy", "pcsf_score_sites_fixed", "pcsf_score_sites_fixed_dev", "pcsf_score_blocks",
                 "pcsf_score_alignments_6frames", "pcsf_score_alignments_6frames_bls", "pcsf_set_bls_branch_lengths",
                 "pcsf_branch_length_scores", "pcsf_block_loglik", "pcsf_host_alloc", "pcsf_host_free", "pcsf_last_error"):
        assert must in names, must
    assert len(names) >= 15


def test_library_exports_every_declared_symbol():
    import phylocsfpp_amd as pa
    lib_path = os.environ.get("PCSF_LIBRARY", os.path.join(ROOT, "phylocsfpp_amd", "libphylocsf_amd.so"))
    assert os.path.exists(lib_path), "build first: python -m phylocsfpp_amd._build"
    lib = ctypes.CDLL(lib_path)
    missing = [n for n in _declared_functions() if not hasattr(lib, n)]
    assert not missing, missing
    assert pa.load_library() is not None
    lib.pcsf_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.pcsf_version()      # host-only call: no device is touched


def test_wig_text_estimate_needs_no_device():
    """pcsf_wig_text_estimate: a host-side bound the caller sizes its text buffer with before pcsf_score_alignments_6frames_wig"""
    import ctypes as C
    import phylocsfpp_amd as pa
    lib = C.CDLL(pa._LIB_PATH)
    lib.pcsf_wig_text_estimate.argtypes = [C.c_size_t, C.c_size_t]
    lib.pcsf_wig_text_estimate.restype = C.c_size_t
    small, big = lib.pcsf_wig_text_estimate(1000, 1), lib.pcsf_wig_text_estimate(1_000_000, 120)
    # at least 12 bytes per possible line (six frames of n / 3 codons and the power track's n / 3) and a header per file and alignment
    assert small >= 12 * (2000 + 333) + 7 * 60 and big >= 12 * (2_000_000 + 333_333) + 7 * 120 * 60 and big < 40_000_000
