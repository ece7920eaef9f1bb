#!/bin/bash
# A variant of the library with one source recompiled under extra flags:  tools/build_variant.sh NAME source.hip -DFLAG ...
#   -> tools/_build/lib_NAME.so (use it with PCSF_LIBRARY=...; the default build must exist)
cd "$(dirname "$0")/.."
NAME=$1; SRC=$2; shift 2
B=phylocsfpp_amd/_build
mkdir -p tools/_build
EXTRA=""
case $SRC in kernels.hip|factored.hip|tiles.hip) EXTRA="-mllvm -structurizecfg-skip-uniform-regions=1";; esac
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -ffp-contract=off $EXTRA "$@" -x hip -c phylocsfpp_amd/csrc/$SRC -o tools/_build/$NAME.$SRC.o || exit 1
OBJS=""
for o in $B/*.o; do [ "$(basename $o)" = "$SRC.o" ] && OBJS="$OBJS tools/_build/$NAME.$SRC.o" || OBJS="$OBJS $o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_build/lib_$NAME.so $OBJS && echo tools/_build/lib_$NAME.so
