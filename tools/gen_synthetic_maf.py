#!/usr/bin/env python3
"""Synthetic multiz-style MAF for BASELINE.json configs[2].

  python tools/gen_synthetic_maf.py OUT.maf [--bases 1000000] [--seed 20261002] [--tree 100vertebrates]
                                    [--exons OUT.exons.txt]

Species = the leaf labels of the 100vertebrates tree written as <label>.chrS (the 58mammals labels are a
subset; the other rows are skipped by a 58mammals run exactly like in `build-tracks 58mammals hg38.100way.maf`).
Reference row first, '+' strand.  Blocks are contiguous on the reference, ~Geom(mean 200) columns, 2 % of
the block boundaries skip 50-500 reference bases (so the reference's unit of parallel work, the concatenated
alignment, is ~100 per Mb).  Sequences are evolved down the tree codon-wise: 10 % coding segments (coding
ECM), the rest with the non-coding ECM; per-block species drop-out 0.3 (not for the first 8 leaves), gap runs,
3 % reference-gap columns, 20 % lower case.
--exons writes the coding segments as the exon list `--coding-exons` reads (chrom strand frame start end, 1-based
inclusive; reference src/estimate_hmm_parameter.hpp:262-283) for BASELINE.json configs[4].
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


SEGMENT_CODONS = 120


def generate(out, bases=1_000_000, seed=20261002, tree=None, start=1_000_000, exons=None, clade_dropout=False):
    """Writes the MAF (and, optionally, the matching coding-exon list); returns a dict describing it."""
    import phylocsfpp_amd as pa
    from phylocsfpp_amd import synth
    m = pa.load_model(tree or "100vertebrates")   # a shipped model, by name
    rng = np.random.default_rng(seed)
    n_cod = (bases + 2) // 3 + 8
    cod, is_coding = synth.synthetic_codon_columns(m.child1, m.child2, m.branch_len, m.S_c, m.pi_c, m.S_nc, m.pi_nc, n_cod,
                                                   seed=seed, dropout=0.0, gap_rate=0.0, segment_codons=SEGMENT_CODONS,
                                                   return_segments=True)
    # codons -> nucleotides [nl, 3*n_cod]
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    nuc = np.empty((m.nl, 3 * n_cod), dtype=np.uint8)
    c = cod.astype(np.int64)
    nuc[:, 0::3] = lut[(c >> 4) & 3]
    nuc[:, 1::3] = lut[(c >> 2) & 3]
    nuc[:, 2::3] = lut[c & 3]
    nuc = nuc[:, :bases]
    lower = rng.random(nuc.shape) < 0.20
    nuc = np.where(lower, nuc | 0x20, nuc)
    labels = m.labels[:m.nl]
    chrom_len = 2 * (start + bases + 100_000)
    ref_of = np.empty(bases, dtype=np.int64)   # 0-based reference position of generated base k (blocks may skip bases)
    n_blocks = 0
    # --clade-dropout: species absence as real multiz alignments show it -- by clade and in long stretches.  Along the genome, stretches
    # of ~Geom(mean 25) blocks keep only the species below one ancestor of the reference species (picked at random from the path to
    # the root, deep ancestors = everything) before the per-block drop-out.  Uses a generator of its own: the default file is unchanged.
    rng_clade = np.random.default_rng(seed + 1)
    parent = np.asarray(m.parent)
    path = []
    v = 0
    while parent[v] >= 0:
        v = int(parent[v])
        path.append(v)
    c1, c2 = np.asarray(m.child1), np.asarray(m.child2)

    def leaves_below(node):
        out_, stack = [], [node]
        while stack:
            u = stack.pop()
            if u < m.nl:
                out_.append(u)
            else:
                stack += [int(c1[u]), int(c2[u])]
        return out_
    stretch_left, stretch_keep = 0, np.ones(m.nl, bool)
    with open(out, "w") as f:
        f.write("##maf version=1 scoring=synthetic\n")
        pos, ref = 0, start
        while pos < bases:
            L = int(min(rng.geometric(1.0 / 200.0), bases - pos))
            present = rng.random(m.nl) >= 0.3
            present[:8] = True
            if clade_dropout:
                if stretch_left == 0:
                    stretch_left = int(rng_clade.geometric(1.0 / 25.0))
                    anc = path[min(len(path) - 1, int(rng_clade.geometric(0.12)) - 1 + 2)]
                    stretch_keep = np.zeros(m.nl, bool)
                    stretch_keep[leaves_below(anc)] = True
                stretch_left -= 1
                present &= stretch_keep
                present[0] = True
            # reference-gap columns: '-' in the reference row, random bases in the other rows, BEFORE base k when ins[k]
            ins = rng.random(L) < 0.03
            cols = L + int(ins.sum())
            base_col = np.arange(L) + np.cumsum(ins)
            ins_col = base_col[ins] - 1
            f.write("a score=%d.000000\n" % (1000 + L))
            for l in range(m.nl):
                if not present[l]:
                    continue
                row = nuc[l, pos:pos + L]
                if len(ins_col):
                    wide = np.empty(cols, dtype=np.uint8)
                    wide[base_col] = row
                    wide[ins_col] = ord("-") if l == 0 else lut[rng.integers(0, 4, size=len(ins_col))]
                    row = wide
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
            ref_of[pos:pos + L] = np.arange(ref, ref + L)
            pos += L
            ref += L
            n_blocks += 1
            if rng.random() < 0.02:
                ref += int(rng.integers(50, 501))
    n_exons = 0
    if exons:
        with open(exons, "w") as f:
            for sgm in np.nonzero(is_coding)[0]:
                b0, b1 = int(sgm) * 3 * SEGMENT_CODONS, min(bases, (int(sgm) + 1) * 3 * SEGMENT_CODONS) - 1
                if b0 >= bases:
                    break
                p0, p1 = int(ref_of[b0]) + 1, int(ref_of[b1]) + 1   # 1-based inclusive
                f.write("chrS\t+\t%d\t%d\t%d\n" % ((p0 - 1) % 3, p0, p1))
                n_exons += 1
    return {"path": out, "bases": bases, "rows": m.nl, "blocks": n_blocks, "ref_start": start, "ref_end": ref,
            "chrom_len": chrom_len, "exons": n_exons, "seed": seed, "bytes": os.path.getsize(out)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--bases", type=int, default=1_000_000)
    ap.add_argument("--seed", type=int, default=20261002)
    ap.add_argument("--tree", default=None, help="model whose tree names the species (default: the shipped 100vertebrates)")
    ap.add_argument("--start", type=int, default=1_000_000, help="0-based reference start of the first block")
    ap.add_argument("--exons", default=None, help="also write the coding segments as a --coding-exons list")
    ap.add_argument("--clade-dropout", action="store_true", help="species absent by clade and in long stretches, like real multiz alignments")
    args = ap.parse_args()
    info = generate(args.out, args.bases, args.seed, args.tree, args.start, args.exons, args.clade_dropout)
    print("wrote %s: %d reference bases, %d rows, %d blocks%s" % (args.out, info["bases"], info["rows"], info["blocks"],
          (", %d exons -> %s" % (info["exons"], args.exons)) if args.exons else ""))


if __name__ == "__main__":
    main()
