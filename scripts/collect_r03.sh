#!/bin/bash
# Everything profiles/r03_* holds, in one GPU session:  scripts/collect_r03.sh  (outputs under gpurun_out/r03/)
cd "$(dirname "$0")/.."
O=gpurun_out/r03
mkdir -p $O
python bench.py > $O/bench_line.json 2> $O/bench_line.err
python bench.py --cycle W --no-strong > $O/wcycle_bench_line.json 2> $O/wcycle_bench_line.err
MG_CYCLE_FORK=0 python bench.py --cycle W --no-strong --no-cpu > $O/wcycle_serial_bench_line.json 2> /dev/null
python bench.py --mixed --refine 2 --n 32768 --steps 10 --warmup 10 --no-strong --no-cpu > $O/config5_bench_line.json 2> $O/config5.err
python bench.py --mixed --no-cpu > $O/mixed_bench_line.json 2> /dev/null
python bench.py --mode graph --no-cpu --no-strong > $O/graph_bench_line.json 2> /dev/null
python scripts/perf_levels.py 8192 > $O/vcycle_levels.txt 2>&1
CYCLE=W python scripts/perf_levels.py 8192 > $O/wcycle_levels.txt 2>&1
python scripts/perf_levels.py 16384 > $O/vcycle_levels_16384.txt 2>&1
python scripts/perf_slab.py 16384 8 > $O/slab8_16384.txt 2>&1
python scripts/perf_slab.py 23040 8 > $O/slab8_23040.txt 2>&1
for w in 1 3 10 30; do WARM=$w TRIALS=6 python scripts/perf_window.py 8192 V eager 20 2>&1 | grep -v amdgpu; done > $O/warmup.txt
MG_LIB=variants/libmg_tiletrace.so python scripts/trace_tile.py 1024 2> $O/tile_trace_raw.txt
grep "tile trace" $O/tile_trace_raw.txt | tail -8 > $O/tile_trace.txt
scripts/profile.sh r03 > $O/profile_tail.txt 2>&1
echo collected
