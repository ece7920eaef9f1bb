cd /root/repo
for q in 4 8; do for g in 1 2 3; do
  echo "== GPU_MAX_HW_QUEUES=$q PCSF_BLOCK_GROUPS=$g"
  GPU_MAX_HW_QUEUES=$q PCSF_BLOCK_GROUPS=$g python3 tools/bench_blocks.py - 0 1 nocpu 2>&1 | grep "GPU OMEGA\|GPU MLE"
  GPU_MAX_HW_QUEUES=$q PCSF_BLOCK_GROUPS=$g python3 tools/bench_blocks.py - 0 1 nocpu 2>&1 | grep "GPU OMEGA\|GPU MLE"
done; done
