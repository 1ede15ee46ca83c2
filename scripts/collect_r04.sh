#!/bin/bash
# Everything profiles/r04_* holds, in one GPU session:  scripts/collect_r04.sh  (outputs under gpurun_out/r04/)
cd "$(dirname "$0")/.."
O=gpurun_out/r04
mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_line_driver_flags.json 2> $O/bench_line_driver_flags.err   # the driver's command
python bench.py > $O/bench_line.json 2> $O/bench_line.err                                                     # 50 + 50 windows
python bench.py --cycle W --no-strong > $O/wcycle_bench_line.json 2> $O/wcycle_bench_line.err
MG_CYCLE_BATCH=0 python bench.py --cycle W --no-strong --no-cpu --steps 10 --warmup 5 > $O/wcycle_serial_bench_line.json 2> /dev/null
python bench.py --cycle W --mode graph --no-strong --no-cpu > $O/wcycle_graph_bench_line.json 2> /dev/null
python bench.py --mixed --refine 2 --n 32768 --steps 10 --warmup 10 --no-strong --no-cpu > $O/config5_bench_line.json 2> $O/config5.err
python bench.py --mixed --no-cpu > $O/mixed_bench_line.json 2> /dev/null
python bench.py --n 4096 --no-cpu --no-strong > $O/config2_bench_line.json 2> /dev/null
python scripts/perf_levels.py 8192 > $O/vcycle_levels.txt 2>&1
CYCLE=W python scripts/perf_levels.py 8192 > $O/wcycle_levels.txt 2>&1
MG_CYCLE_DEBUG=1 python scripts/perf_window.py 8192 W eager 5 2>&1 | grep -E "cycle\]|depth|W\(" > $O/wcycle_schedule.txt
python scripts/perf_levels.py 16384 > $O/vcycle_levels_16384.txt 2>&1
python scripts/perf_slab.py 16384 8 > $O/slab8_16384.txt 2>&1
python scripts/perf_slab.py 8192 8 > $O/slab8_8192.txt 2>&1
python scripts/perf_slab.py 23040 8 > $O/slab8_23040.txt 2>&1
python scripts/perf_slab.py 32768 8 > $O/slab8_32768.txt 2>&1
scripts/profile.sh r04 > $O/profile_tail.txt 2>&1
echo collected
