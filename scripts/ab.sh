#!/bin/bash
# A/B of two builds of the library in ONE GPU session (boxes of the pool differ by several per cent):
#   scripts/ab.sh <libA.so> <libB.so> [sizes...]   -> per-node kernel times and whole-cycle times, interleaved
A=$1; B=$2; shift 2
SIZES=${@:-"8192 4096 2048 1024 512 256 128"}
for n in $SIZES; do
  for rep in 1 2; do
    TAG=A MG_LIB=$A python3 scripts/perf_nodes.py $n 2>/dev/null | grep -E "prolong|restrict"
    TAG=B MG_LIB=$B python3 scripts/perf_nodes.py $n 2>/dev/null | grep -E "prolong|restrict"
  done
done
for rep in 1 2; do
  echo "A: $(MG_LIB=$A python3 scripts/perf_levels.py 8192 2>/dev/null | head -1)"
  echo "B: $(MG_LIB=$B python3 scripts/perf_levels.py 8192 2>/dev/null | head -1)"
done
