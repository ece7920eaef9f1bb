#!/bin/bash
# kernel timeline of bench_blocks.py on the reference's 516-block file (env passed through): tools/blocks516_timeline.sh [replicas]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf /tmp/tl516
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl516 -- python3 tools/bench_blocks.py - 0 ${1:-1} nocpu 2>&1 | grep GPU
python3 tools/kernel_timeline.py /tmp/tl516 eigen_omega
