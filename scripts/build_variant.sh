#!/bin/bash
# A second build of the library with extra compiler flags (diagnostics such as -DMG_TAIL_PHASES or
# -DMG_STREAM_TRACE, or an experiment), next to the product build; select it with MG_LIB=<out.so>.
#   scripts/build_variant.sh <out.so> <flags...>
set -e
OUT=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/multigrid_poisson_solver_amd/csrc
TMP=$(mktemp -d)
mkdir -p "$(dirname "$OUT")"
for s in $CSRC/*.hip $CSRC/mg_abi.cpp $CSRC/mg_tables.cpp $CSRC/mg_cycle.cpp $CSRC/mg_slab.cpp $CSRC/mg_comm.cpp; do
  x=""; case $s in *.hip) x="-x hip";; esac
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I$ROOT/include -I$CSRC -Wno-unused-result "$@" $x -c $s -o $TMP/$(basename $s).o &
done
wait
hipcc --offload-arch=gfx950 -shared -o $OUT $TMP/*.o -lpthread -ldl
rm -rf $TMP
echo $OUT
