#!/bin/bash
# W-cycle at 8192: forked sub-cycles -- streams, fork threshold, hardware queues, graph replay
cd "$(dirname "$0")/.."
for q in "" 8 16; do
  for st in 8 16 32; do
    for mx in 1024 4096; do
      if [ -n "$q" ]; then export GPU_MAX_HW_QUEUES=$q; else unset GPU_MAX_HW_QUEUES; fi
      MG_FORK_STREAMS=$st MG_FORK_MAX_N=$mx python scripts/perf_window.py 8192 W eager 10 2>&1 | grep -v amdgpu
    done
  done
done
unset GPU_MAX_HW_QUEUES
MG_CYCLE_FORK=0 python scripts/perf_window.py 8192 W eager 10 2>&1 | grep -v amdgpu
python scripts/perf_window.py 8192 W graph 10 2>&1 | grep -v amdgpu
MG_FORK_STREAMS=16 MG_FORK_MAX_N=4096 python scripts/perf_window.py 8192 W graph 10 2>&1 | grep -v amdgpu
MG_CYCLE_FORK=0 python scripts/perf_window.py 8192 W graph 10 2>&1 | grep -v amdgpu
python scripts/perf_window.py 8192 V eager 20 2>&1 | grep -v amdgpu
python scripts/perf_window.py 8192 V graph 20 2>&1 | grep -v amdgpu
