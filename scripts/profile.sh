#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel trace + HBM counters of bench.py.
#   scripts/profile.sh <tag> [bench args...]
# Writes raw output under gpurun_out/prof_<tag>/ and a summary gpurun_out/prof_<tag>/summary.md
# (copy the summary into profiles/ to have it judged).
set -e
TAG=${1:-r01}; shift || true
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# the profiler's tool library belongs to the ROCm installation: bind the engine to that HIP runtime, not to the copy
# a PyTorch wheel bundles (multigrid_poisson_solver_amd/__init__.py:_bind_hip_runtime)
export MG_HIP_RUNTIME=system
ARGS="--no-cpu --no-strong $@"   # bench.py's own K and W (50 + 50 windows back to back): the traced durations are those of the steady state the bench line reports
# 1. kernel trace + stats (no counters)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $ROOT/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.log
# 2./3. HBM counters, one pass each (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- python3 $ROOT/bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- python3 $ROOT/bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.log
python3 $ROOT/scripts/summarize_profile.py $OUT $TAG > $OUT/summary.md
tail -40 $OUT/summary.md
