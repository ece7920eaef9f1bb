#!/bin/bash
# A/B of alternative builds of the library: tools/lib_ab.sh lib1.so lib2.so ...   (default build first)
cd "$(dirname "$0")/.."
echo "== default"; python tools/ab_bench.py 2000000 58mammals v0 2>&1 | tail -1
for l in "$@"; do echo "== $l"; PCSF_LIBRARY=$PWD/$l python tools/ab_bench.py 2000000 58mammals v0 2>&1 | tail -1; done
