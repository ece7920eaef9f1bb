#!/bin/bash
# one gpurun call: the whole GPU suite, the block-strategy benchmark, the headline bench (outputs under gpurun_out/$1)
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
timeout ${PYTEST_TIMEOUT:-2400} python -m pytest tests -m gpu -q -s ${PYTEST_ARGS:-} > $out/pytest.txt 2>&1
echo "pytest rc=$?" >> $out/pytest.txt
tail -n 60 $out/pytest.txt
timeout 900 python tools/bench_blocks.py > $out/blocks_bench_516.txt 2>&1
cat $out/blocks_bench_516.txt
if [ -z "$SKIP_BENCH" ]; then
  timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
  echo "bench rc=$?"; tail -c 6000 $out/bench.json; tail -n 5 $out/bench.err
fi
