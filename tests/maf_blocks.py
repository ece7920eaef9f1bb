"""Test helper: single-block (score-msa style) codon matrices from a MAF file, in pure Python.
Independent of the C++ host driver; used to feed the C-ABI and the oracle with identical inputs."""
import gzip
import os

import numpy as np

DNA = {"A": 0, "a": 0, "C": 1, "c": 1, "G": 2, "g": 2, "T": 3, "t": 3}


def load_aliases(path):
    alias = {}
    for line in open(path):
        common, names = line.rstrip("\n").split("\t")
        for x in names.split():
            alias[x.lower()] = common
    return alias


def block_codons(maf_path, labels, alias_path, max_blocks=None):
    """returns (codons uint8 [nl, total], offsets uint64 [n_blocks+1])"""
    alias = load_aliases(alias_path)
    lab2id = {l: i for i, l in enumerate(labels) if l}
    nl = sum(1 for l in labels if l)

    def ident(s):
        s = s.lower()
        if s in lab2id:
            return lab2id[s]
        return lab2id.get(alias.get(s))

    opener = gzip.open if maf_path.endswith(".gz") else open
    blocks, cur = [], None
    with opener(maf_path, "rt") as f:
        for line in f:
            if line.startswith("a"):
                cur = []
                blocks.append(cur)
            elif line.startswith("s") and cur is not None:
                cur.append(line.split())
    if max_blocks:
        blocks = blocks[:max_blocks]
    mats, offs = [], [0]
    for b in blocks:
        seqs, ref = [None] * nl, None
        for f in b:
            i = ident(f[1].split(".")[0])
            if i is None:
                continue
            if ref is None:
                ref = i
            seqs[i] = f[6]
        L = len(seqs[ref])
        seqs = [s if s is not None else "N" * L for s in seqs]
        keep = [k for k in range(L) if seqs[ref][k] != "-"]
        nc = len(keep) // 3
        cod = np.full((nl, nc), 64, np.uint8)
        for i, s in enumerate(seqs):
            s = "".join(s[k] for k in keep)
            for c in range(nc):
                t = s[3 * c:3 * c + 3]
                if all(ch in DNA for ch in t):
                    cod[i, c] = 16 * DNA[t[0]] + 4 * DNA[t[1]] + DNA[t[2]]
        mats.append(cod)
        offs.append(offs[-1] + nc)
    total = offs[-1]
    out = np.full((nl, max(total, 1)), 64, np.uint8)
    for m, o in zip(mats, offs):
        out[:, o:o + m.shape[1]] = m
    return out, np.array(offs, dtype=np.uint64)


def block_rows(maf_path, labels, alias_path, max_blocks=None):
    """per block: the ASCII rows uint8 [nl, L] with the reference row's gap columns removed and absent species
    filled with N (what score-msa hands to compute_bls_score)"""
    alias = load_aliases(alias_path)
    lab2id = {l: i for i, l in enumerate(labels) if l}
    nl = sum(1 for l in labels if l)

    def ident(s):
        s = s.lower()
        if s in lab2id:
            return lab2id[s]
        return lab2id.get(alias.get(s))

    opener = gzip.open if maf_path.endswith(".gz") else open
    blocks, cur = [], None
    with opener(maf_path, "rt") as f:
        for line in f:
            if line.startswith("a"):
                cur = []
                blocks.append(cur)
            elif line.startswith("s") and cur is not None:
                cur.append(line.split())
    if max_blocks:
        blocks = blocks[:max_blocks]
    out = []
    for b in blocks:
        seqs, ref = [None] * nl, None
        for f in b:
            i = ident(f[1].split(".")[0])
            if i is None:
                continue
            if ref is None:
                ref = i
            seqs[i] = f[6]
        L = len(seqs[ref])
        keep = np.array([k for k in range(L) if seqs[ref][k] != "-"], dtype=np.int64)
        mat = np.full((nl, len(keep)), ord("N"), np.uint8)
        for i, s in enumerate(seqs):
            if s is not None:
                mat[i] = np.frombuffer(s.encode(), dtype=np.uint8)[keep]
        out.append(mat)
    return out
