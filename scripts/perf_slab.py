#!/usr/bin/env python3
"""Per-launch breakdown of the row-slab driver with R virtual ranks on one GPU (R = 1: the pure overhead of
the slab code path against the single-GPU driver; R > 1: all slabs run one after the other on this GPU)."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multigrid_poisson_solver_amd as mg

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1
reps = 10
mg.init(0)
path = os.path.join(tempfile.mkdtemp(), "v.txt")
mg.write_vcycle_file(path, N, 8, 3, 1e-7)
plan = mg.SlabPlan(path, R, -1, int(os.environ.get("MG_COLLAPSE_N", "1024")))
plan.want_error(False)
for _ in range(3):
    plan.execute()
for _ in range(int(os.environ.get("WARM", "40")) // max(1, R if R > 1 else 1) + 4):   # sustained load first: the clocks ramp for ~30 ms
    plan.enqueue()
mg.sync()
plan.collect()
t0 = time.perf_counter()
for _ in range(reps):
    plan.enqueue()
mg.sync()
t1 = time.perf_counter()
plan.collect()
print(f"slab window (R={R}): {(t1 - t0) / reps * 1e3:.4f} ms per cycle")
for _ in range(4):
    plan.enqueue()
mg.sync()
plan.collect()
mg.profile_begin(0)
for _ in range(reps):
    plan.enqueue()
mg.sync()
plan.collect()
tot = 0.0
for e in sorted(mg.profile_end(), key=lambda e: (-e["N"], e["name"])):
    avg = e["total_ms"] / reps
    tot += avg
    print(f"N={e['N']:>5} {e['name']:<40} {e['launches'] // reps:>3} x {e['total_ms'] / e['launches'] * 1e3:8.1f} us = {avg * 1e3:8.1f} us/cycle")
print(f"sum of profiled launches: {tot:.4f} ms per cycle")
