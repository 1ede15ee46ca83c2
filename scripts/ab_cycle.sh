#!/bin/bash
# A/B of two builds on the V-cycle at one size, interleaved in ONE GPU session:  scripts/ab_cycle.sh <libA.so> <libB.so> [N]
A=$1; B=$2; N=${3:-8192}
for rep in 1 2 3; do
  for L in A B; do
    lib=$A; [ $L = B ] && lib=$B
    echo "$L: $(MG_LIB=$lib python3 scripts/perf_window.py $N V eager 20 2>/dev/null | grep -v amdgpu | cut -c1-60)"
    MG_LIB=$lib python3 scripts/perf_levels.py $N 2>/dev/null | grep -E "N= *($N|$((N/2))) " | sed "s/^/$L:   /"
  done
done
