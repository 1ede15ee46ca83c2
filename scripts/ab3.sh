#!/bin/bash
# A/B/C... of several builds on the V-cycle at one size, interleaved in ONE GPU session:  scripts/ab3.sh N lib1.so lib2.so ...
N=$1; shift
for rep in 1 2 3; do
  for lib in "$@"; do
    tag=$(basename $lib .so)
    echo "$tag: $(MG_LIB=$lib python3 scripts/perf_window.py $N V eager 20 2>/dev/null | grep -v amdgpu | cut -c1-60)"
    MG_LIB=$lib python3 scripts/perf_levels.py $N 2>/dev/null | grep -E "N= *($N|$((N/2))) " | sed "s/^/$tag:   /"
  done
done
