#!/bin/bash
# FORM 2 of the eigen-solver against the product's form: accuracy test, launch durations at 200 and 2,048 jobs, in-kernel stamps of the step-stamps build
cd "$(dirname "$0")/.." || exit 1
python -m pytest tests/test_gpu_blocks.py -x -q -k "eigen" 2>&1 | tail -3
FORMS="reg quad" bash tools/eigen_ab.sh
EIGEN_JOBS=2048 FORMS="reg quad" bash tools/eigen_ab.sh | grep "form\|launches"
[ -f tools/_build/lib_f2stamps.so ] && PCSF_EIGEN_FORM=2 PCSF_EIGEN_STAMPS=1 PCSF_LIBRARY=$PWD/tools/_build/lib_f2stamps.so python3 tools/eigen_bench.py 200 2>&1 | grep "eigen stamps" | sed -n '2p;4p'
[ -f tools/_build/lib_f2rstamps.so ] && echo "re-seatings (per step: staging / barrier / collecting / rotations + hand-over):" && PCSF_EIGEN_FORM=2 PCSF_EIGEN_STAMPS=1 PCSF_LIBRARY=$PWD/tools/_build/lib_f2rstamps.so python3 tools/eigen_bench.py 200 2>&1 | grep "eigen stamps" | sed -n '2p;4p'
for v in $F2_VARIANTS; do echo "== variant $v"; PCSF_EIGEN_FORM=2 FORMS=none bash tools/eigen_ab.sh tools/_build/lib_$v.so | grep "stamps\|launches" | sed -n '2p;4p'; done
