#!/bin/bash
# kernel time of tools/tile_bench.py: tools/tile_prof.sh [tile_bench args]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf /tmp/tile_prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tile_prof -- python3 tools/tile_bench.py "$@" 2>&1 | grep blocks
grep -E "prune_tiles|prune_factored|edge_scale" $(find /tmp/tile_prof -name "*kernel_stats.csv" | head -1) | cut -c1-140
