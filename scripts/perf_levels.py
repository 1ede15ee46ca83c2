#!/usr/bin/env python3
"""Per-level breakdown of one V(nu,nu)-cycle: live hipEvent pairs around every launch of the
fused driver, summed per (kernel, N), next to the whole-window time.  Shows where the time
below the finest level goes (small levels are latency-, not bandwidth-bound)."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multigrid_poisson_solver_amd as mg

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = 5 if os.environ.get("CYCLE") == "W" else 20
mg.init(0)
path = os.path.join(tempfile.mkdtemp(), "v.txt")
(mg.write_wcycle_file if os.environ.get("CYCLE") == "W" else mg.write_vcycle_file)(path, N, 8, 3, 1e-7)
plan = mg.CyclePlan(path, fused=True, report=False, error=False, mixed=bool(os.environ.get("MIXED")))
for _ in range(3):
    plan.execute()
mg.sync()
t0 = time.perf_counter()
for _ in range(reps):
    plan.enqueue()
mg.sync()
t1 = time.perf_counter()
plan.collect()
print(f"window: {(t1 - t0) / reps * 1e3:.4f} ms per cycle (no profiling events)")
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
