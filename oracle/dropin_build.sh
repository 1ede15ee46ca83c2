#!/bin/bash
# oracle/dropin_build.sh -- TEST INFRASTRUCTURE.  Builds the REFERENCE's own driver
# (main() of src/MG_solver_CPU.cpp, its linked list) against libmgpoisson.so, applying exactly
# the maintainer-side edits INTEGRATION.md describes.  The patched sources exist only in a
# temporary directory; the only output is the binary oracle/_ref/MG_HIP_dropin (git-ignored,
# shipped to the GPU box like the other oracle/_ref artefacts).  tests/test_cycle_gpu.py runs it
# on the shipped cycle files: the reference's main() drives the HIP engine through
# include/mg_dropin.hpp and must print what the reference program printed.
set -e
REF=${REF:-/root/reference/src}
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(dirname "$HERE")
LIBDIR=$ROOT/multigrid_poisson_solver_amd/lib
[ -f "$REF/MG_solver_CPU.cpp" ] || { echo "reference tree not present: keeping prebuilt oracle/_ref/MG_HIP_dropin if any"; exit 0; }
[ -f "$LIBDIR/libmgpoisson.so" ] || { echo "build libmgpoisson.so first"; exit 1; }
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT

# main(): lines 1-462 of the reference file (operators are defined below it, :464-1068)
head -n 462 "$REF/MG_solver_CPU.cpp" | sed \
  -e '16,34d' \
  -e '36a\    if (mg_init(0) != 0) return 1;   /* replaces cudaSetDevice(0), MG_solver_GPU.cu:58 */' \
  -e '213s/.*/\t\t\t\t\tmg_fill_zero(U, (size_t)N * N);/' \
  -e '256s/.*/\t\t\t\t\tmg_fill_zero(U, (size_t)N * N);/' \
  -e '277,280c\				mg_negate(N, D);' \
  -e '353s/.*/\t\t\ttempU = mg_alloc((size_t)next_N * next_N);/' \
  -e '371s/.*/\t\t\tmg_free(tempU);/' \
  -e '438,445c\	double MGerror = 0.0; mg_analyticError(N, L, U, min_x, min_y, \&MGerror);' \
  > "$TMP/body.cpp"
{ echo '#include "mg_dropin.hpp"'; cat "$TMP/body.cpp"; } > "$TMP/MG_solver_HIP.cpp"

sed -e '1i #include "mg_hip.h"' \
    -e 's/(double\*) malloc(N \* N \* sizeof(double))/mg_alloc((size_t)N * N)/' \
    -e 's/free(\(.* -> [UFD]\));/mg_free(\1);/' \
    "$REF/linkedlist.cpp" > "$TMP/linkedlist_hip.cpp"
cp "$REF/linkedlist.h" "$TMP/"

mkdir -p "$HERE/_ref"
g++ -O2 -fopenmp -w -I"$ROOT/include" -I"$TMP" -o "$HERE/_ref/MG_HIP_dropin" \
    "$TMP/MG_solver_HIP.cpp" "$TMP/linkedlist_hip.cpp" \
    -L"$LIBDIR" -lmgpoisson -Wl,-rpath,'$ORIGIN/../../multigrid_poisson_solver_amd/lib'
echo "built $HERE/_ref/MG_HIP_dropin"
