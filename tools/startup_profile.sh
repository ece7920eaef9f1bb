#!/bin/bash
# Where the wall time of one `phylocsf-amd build-tracks` process on the 1 Mb synthetic MAF goes: the process as the shell sees it,
# the CLI's own start-up line, and the floor of any HIP process on the box (tools/hip_init_time.cpp).
#   bash tools/startup_profile.sh  > gpurun_out/startup_profile.txt
set -u
cd "$(dirname "$0")/.."
W=${TMPDIR:-/tmp}/pcsf_startup; mkdir -p $W
python tools/gen_synthetic_maf.py $W/s.maf --bases 1000000 > /dev/null
# the floor program is built where it runs (tools/_build/ is not part of the tree)
mkdir -p tools/_build
[ -x tools/_build/hip_init_time ] || hipcc -O2 --offload-arch=gfx950 -o tools/_build/hip_init_time tools/hip_init_time.cpp > /dev/null 2>&1
echo "== floor: a HIP process that initialises the runtime, makes a context, one allocation, one stream (three runs)"
for k in 1 2 3; do { TIMEFORMAT="   process wall %R s (user %U, sys %S)"; time tools/_build/hip_init_time; } 2>&1; done
echo "== phylocsf-amd build-tracks 58mammals <1 Mb, 90 MB MAF>, five runs"
for k in 1 2 3 4 5; do
  { TIMEFORMAT="process wall %R s (user %U, sys %S)"; time phylocsfpp_amd/host/phylocsf-amd build-tracks --output $W/o tests/golden/models/58mammals $W/s.maf; } 2>&1 | grep -v "^$" | sed 's/^/   /'
done
echo "== the same with one scoring context"
for k in 1 2; do
  { TIMEFORMAT="process wall %R s (user %U, sys %S)"; time phylocsfpp_amd/host/phylocsf-amd build-tracks --contexts-per-device 1 --output $W/o tests/golden/models/58mammals $W/s.maf; } 2>&1 | grep -v "^$" | sed 's/^/   /'
done
