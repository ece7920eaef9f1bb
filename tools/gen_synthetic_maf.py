#!/usr/bin/env python3
"""Synthetic multiz-style MAF for BASELINE.json configs[2].

  python tools/gen_synthetic_maf.py OUT.maf [--bases 1000000] [--seed 20261002] [--tree tests/golden/models/100vertebrates]

Species = the leaf labels of the 100vertebrates tree written as <label>.chrS (the 58mammals labels are a
subset; the other rows are skipped by a 58mammals run exactly like in `build-tracks 58mammals hg38.100way.maf`).
Reference row first, '+' strand.  Blocks are contiguous on the reference, ~Geom(mean 200) columns, 2 % of
the block boundaries skip 50-500 reference bases (so the reference's unit of parallel work, the concatenated
alignment, is ~100 per Mb).  Sequences are evolved down the tree codon-wise: 10 % coding segments (coding
ECM), the rest with the non-coding ECM; per-block species drop-out 0.3 (not for the first 8 leaves), gap runs,
3 % reference-gap columns, 20 % lower case.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--bases", type=int, default=1_000_000)
    ap.add_argument("--seed", type=int, default=20261002)
    ap.add_argument("--tree", default=os.path.join(ROOT, "tests/golden/models/100vertebrates"))
    ap.add_argument("--start", type=int, default=1_000_000, help="0-based reference start of the first block")
    args = ap.parse_args()
    import oracle_binding as ob           # only used as a model-file reader here
    from phylocsfpp_amd import synth
    m = ob.OracleModel(args.tree)
    rng = np.random.default_rng(args.seed)
    n_cod = (args.bases + 2) // 3 + 8
    cod = synth.synthetic_codon_columns(m.child1, m.child2, m.branch_len, m.S_c, m.pi_c, m.S_nc, m.pi_nc, n_cod,
                                        seed=args.seed, dropout=0.0, gap_rate=0.0)
    # codons -> nucleotides [nl, 3*n_cod]
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    nuc = np.empty((m.nl, 3 * n_cod), dtype=np.uint8)
    c = cod.astype(np.int64)
    nuc[:, 0::3] = lut[(c >> 4) & 3]
    nuc[:, 1::3] = lut[(c >> 2) & 3]
    nuc[:, 2::3] = lut[c & 3]
    nuc = nuc[:, :args.bases]
    lower = rng.random(nuc.shape) < 0.20
    nuc = np.where(lower, nuc | 0x20, nuc)
    labels = m.labels[:m.nl]
    chrom_len = 2 * (args.start + args.bases + 100_000)
    with open(args.out, "w") as f:
        f.write("##maf version=1 scoring=synthetic\n")
        pos, ref = 0, args.start
        while pos < args.bases:
            L = int(min(rng.geometric(1.0 / 200.0), args.bases - pos))
            present = rng.random(m.nl) >= 0.3
            present[:8] = True
            # reference-gap columns: insert '-' into the reference row (other rows get random bases there)
            ins = rng.random(L) < 0.03
            cols = L + int(ins.sum())
            f.write("a score=%d.000000\n" % (1000 + L))
            for l in range(m.nl):
                if not present[l]:
                    continue
                row = nuc[l, pos:pos + L]
                if ins.any():
                    out = np.empty(cols, dtype=np.uint8)
                    idx = np.cumsum(np.concatenate([[0], 1 + ins[:-1].astype(np.int64)]))  # ref base k sits at column idx[k] (+1 if an insertion precedes... )
                    idx = np.arange(L) + np.cumsum(ins) - ins.astype(np.int64) + ins.astype(np.int64)
                    # insertion column sits BEFORE base k when ins[k]
                    base_col = np.arange(L) + np.cumsum(ins)
                    ins_col = base_col[ins] - 1
                    out[base_col] = row
                    if l == 0:
                        out[ins_col] = ord("-")
                    else:
                        out[ins_col] = lut[rng.integers(0, 4, size=len(ins_col))]
                    row = out
                if l != 0 and rng.random() < 0.3:           # a gap run in this species
                    row = row.copy()
                    g0 = int(rng.integers(0, len(row)))
                    row[g0:g0 + int(rng.geometric(0.2))] = ord("-")
                if l == 0:
                    f.write("s %s.chrS %d %d + %d %s\n" % (labels[l], ref, L, chrom_len, row.tobytes().decode()))
                else:
                    n_bases = int((row != ord("-")).sum())
                    f.write("s %s.chrS %d %d + %d %s\n" % (labels[l], int(rng.integers(0, 10**7)), n_bases, chrom_len, row.tobytes().decode()))
            f.write("\n")
            pos += L
            ref += L
            if rng.random() < 0.02:
                ref += int(rng.integers(50, 501))
    print("wrote %s: %d reference bases, %d rows" % (args.out, args.bases, m.nl))


if __name__ == "__main__":
    main()
