#!/usr/bin/env python3
"""Writes tests/golden/oracle/{small,medium}_{mle,omega}.npy: the CPU ORACLE's MLE (phylo, anc) and OMEGA (phylo)
block scores on the reference's 50- and 516-block score-msa files, 100vertebrates model, seed 42.

These are NOT reference outputs (those are the *.scores files next door); they are the oracle's own end results, kept
so that the GPU tests can state how many rows agree with the oracle to 1e-6 without spending ~25 core-minutes of
oracle time on the GPU box.  tests/test_oracle_golden.py re-computes a sample of the rows with the live oracle, so a
stale file fails the CPU suite.

    python tests/golden/make_oracle_outputs.py      # ~5 min on 8 cores
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import maf_blocks
import oracle_binding as ob

ALIASES = os.path.join(HERE, "models", "species_aliases.tsv")
FILES = {"small": "chr22.50alignments.maf", "medium": "chr22.516alignments.maf.gz"}


def main():
    m = ob.OracleModel(os.path.join(HERE, "models", "100vertebrates"))
    os.makedirs(os.path.join(HERE, "oracle"), exist_ok=True)
    for tag, name in FILES.items():
        cod, off = maf_blocks.block_codons(os.path.join(HERE, "score-msa", name), m.labels, ALIASES)
        ph, an = m.run_blocks(ob.MLE, cod, off, compute_anc=True)
        np.save(os.path.join(HERE, "oracle", tag + "_mle.npy"), np.stack([ph, an]))
        ph, _ = m.run_blocks(ob.OMEGA, cod, off, compute_anc=False)
        np.save(os.path.join(HERE, "oracle", tag + "_omega.npy"), ph)
        print(tag, "done:", len(off) - 1, "blocks")


if __name__ == "__main__":
    main()
