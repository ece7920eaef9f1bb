#!/usr/bin/env python3
"""Writes the model files that resolve by NAME (phylocsfpp_amd/models/) for ALL eleven models the reference compiles in
(src/models.hpp:1443-1455: 20flies 26worms 12flies 23flies 100vertebrates 49birds 53birds 7yeast 58mammals 21mosquitoes 29mammals), from
the read-only reference checkout (this container only).  What is written is DATA -- the models' parameter tables in the reference's
own file format (get_ecm, src/ecm.hpp:20-67: 63 lower-triangle rows, a blank line, 64 frequencies), their tree strings and the
species-name table (src/models.hpp:1468-1706) -- with every number spelt as the reference spells it, so that strtod yields the double the
reference's compiled-in literal is.  The packed literals list the pairs (i, j), i < j, column by column (src/ecm.hpp:86-103), which is the
file's row order.

  python tests/golden/make_models.py [/root/reference]
"""
import os, re, sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "phylocsfpp_amd", "models")
src = open(os.path.join(REF, "src/models.hpp")).read()
names = re.findall(r'\{\s*"(\w+)",\s*\{\s*&g_\w+_tree', src)
assert len(names) == 11, names


def literal(name):
    m = re.search(r"double\s+%s\[[^\]]*\]\s*=\s*\{([^}]*)\}" % name, src)
    return [x.strip() for x in m.group(1).replace("\n", " ").split(",") if x.strip()]


def write_ecm(path, tri, freq):
    assert len(tri) == 63 * 64 // 2 and len(freq) == 64
    with open(path, "w") as f:
        at = 0
        for j in range(1, 64):
            f.write(" ".join(tri[at:at + j]) + "\n")
            at += j
        f.write("\n" + " ".join(freq) + "\n")


labels = set()
for model in names:
    tree = re.search(r'std::string g_%s_tree\s*=\s*"([^"]*)"' % model, src).group(1)
    with open(os.path.join(OUT, model + ".nh"), "w") as f:
        f.write(tree + "\n")
    labels |= {x.lower() for x in re.findall(r"[(,]\s*([^:,()\s]+)\s*:", tree)}
    write_ecm(os.path.join(OUT, model + "_coding.ECM"), literal("g_%s_cmatrix" % model), literal("g_%s_cfreq" % model))
    write_ecm(os.path.join(OUT, model + "_noncoding.ECM"), literal("g_%s_ncmatrix" % model), literal("g_%s_ncfreq" % model))
block = src[src.index("sequence_name_mapping = {"):]
block = block[:block.index("};")]
rows = re.findall(r'\{\s*"([^"]+)"\s*,\s*\{([^}]*)\}\s*\}', block)
with open(os.path.join(OUT, "species_aliases.tsv"), "w") as f:
    for common, aliases in sorted(rows):
        if common in labels:
            f.write(common + "\t" + " ".join(re.findall(r'"([^"]+)"', aliases)) + "\n")
print("models written to", OUT, ":", " ".join(names))
