#!/bin/bash
# CPU-only: builds the oracle and the host driver with AddressSanitizer + UBSan and runs every CLI mode on small inputs
# (GPU sanitizers are not available on the pool; the host layer and the oracle are plain C/C++).
set -e
R=$(cd "$(dirname "$0")/.." && pwd); W=${1:-/tmp/pcsf_asan}; mkdir -p $W
gcc -O1 -g -fopenmp -fPIC -std=gnu11 -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -shared -o $W/libpcsf_oracle.so $R/oracle/pcsf_oracle.c -lm
g++ -O1 -g -std=c++17 -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer -o $W/oracle_cli $R/oracle/oracle_cli.cpp -L$W -lpcsf_oracle -Wl,-rpath,$W -lm
export ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
python3 - "$R" "$W" <<'PY'
import gzip, sys
R, W = sys.argv[1:3]
t = gzip.open(R + "/tests/golden/build-tracks/galGal6_chr22_25_28_each_30k_bases.maf.gz", "rt").read(3000000)
open(W + "/g.maf", "w").write(t[:t.rfind("\n\na ") + 2])
t = open(R + "/tests/golden/score-msa/chr22.50alignments.maf").read(400000)
open(W + "/s.maf", "w").write(t[:t.rfind("\n\na ") + 2])
PY
H=0.004,55,0.45,0.45,0.1,800,9000,60; M=$R/tests/golden/models
$W/oracle_cli build-tracks --output $W/out --output-phylo 1 --output-regions 1 --hmm-params $H $M/53birds $W/g.maf
$W/oracle_cli build-tracks --output $W/out2 --species Chicken,Turkey,Zebra_finch,Budgerigar $M/53birds $W/g.maf
for st in fixed mle omega fixed_mean; do $W/oracle_cli score-msa --strategy $st --comp-anc 1 --hmm-params $H $M/100vertebrates $W/s.maf; done
$W/oracle_cli dump-alignments --jobs 7 $M/53birds $W/g.maf > /dev/null
$W/oracle_cli pack-maf $M/53birds $W/g.maf $W/g.pcsfbin
$W/oracle_cli build-tracks --output $W/out3 --window-bytes 300000 --batch-sites 5000 --devices 0,0 $M/53birds $W/g.pcsfbin
cmp $W/out2/../out3/PhyloCSFRaw+1.wig $W/out/PhyloCSFRaw+1.wig
head -c 100000 $W/g.pcsfbin > $W/cut.pcsfbin; if $W/oracle_cli build-tracks --output $W/out4 $M/53birds $W/cut.pcsfbin 2>/dev/null; then echo "truncated packed file accepted"; exit 1; fi
printf 'a score=0\ns chicken.chr1 10 4 + 1000 ACXT\ns turkey.chr1 10 4 + 1000 ACGT\n' > $W/bad.maf
if $W/oracle_cli build-tracks --output $W/out5 $M/53birds $W/bad.maf 2>/dev/null; then echo "bad character accepted"; exit 1; fi
test ! -e $W/out5/PhyloCSFRaw+1.wig   # a failed run leaves no truncated outputs behind
$W/oracle_cli smooth-tracks --output-regions 1 --hmm-params $H $W/out
$W/oracle_cli selftest-format 200000
echo "sanitizers: no findings"
