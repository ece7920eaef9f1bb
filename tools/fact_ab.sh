#!/bin/bash
# Runs on the GPU box: prune_factored after a kernel change -- its tests, the cycles of a tile by content (lone work-groups and a
# loaded device), and the block strategies on the reference's real blocks (516, 2,064, 8,256).  bash tools/fact_ab.sh <out dir>
cd ${GRAFT_REPO_ROOT:-.}; O=${1:-gpurun_out/fact_ab}; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_blocks.py tests/test_gpu_parity_blocks.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
{ python3 tools/factored_tile_cost.py 200 2>&1 | grep "stamps"; python3 tools/factored_tile_cost.py 3000 2>&1 | grep stamps; } > $O/factored_tile_cost.txt 2>&1
{ for r in 1 4 16; do echo "== bench_blocks.py - 0 $r nocpu"; PCSF_TIME_BLOCKS=1 python3 tools/bench_blocks.py - 0 $r nocpu 2>&1 | grep -E "GPU |run_omega: [0-9]+ blocks|run_mle: [0-9]+ blocks"; done; } > $O/blocks_real_files.txt 2>&1
cat $O/factored_tile_cost.txt; grep "GPU " $O/blocks_real_files.txt
