#!/usr/bin/env python3
"""Whole-window time of one cycle (no profiling events):  perf_window.py N [V|W] [eager|graph] [reps]
Tuning knobs come from the environment (MG_CYCLE_BATCH, MG_BATCH_TAIL_N, MG_BATCH_RECOMPUTE_MIN_N, ...)."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multigrid_poisson_solver_amd as mg

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
kind = sys.argv[2] if len(sys.argv) > 2 else "V"
mode = sys.argv[3] if len(sys.argv) > 3 else "eager"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
mg.init(0)
path = os.path.join(tempfile.mkdtemp(), "c.txt")
(mg.write_wcycle_file if kind == "W" else mg.write_vcycle_file)(path, N, 8, 3, 1e-7)
plan = mg.CyclePlan(path, fused=True, graph=(mode == "graph"), report=False, error=False)
for _ in range(int(os.environ.get("WARM", "4"))):
    r = plan.execute()
    assert r["status"] == 0, r
best, trials = 1e9, []
for _ in range(int(os.environ.get("TRIALS", "3"))):
    mg.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        plan.enqueue()
    mg.sync()
    trials.append((time.perf_counter() - t0) / reps * 1e3)
    best = min(best, trials[-1])
r = plan.collect()
tags = " ".join(f"{k}={os.environ[k]}" for k in ("MG_CYCLE_BATCH", "MG_BATCH_TAIL_N", "MG_BATCH_RECOMPUTE_MIN_N", "MG_RECOMPUTE_MIN_N") if k in os.environ)
print(f"{kind}({N}) {mode:5s} {best:8.4f} ms per window (best of 3 x {reps}), status {r['status']}  {tags}  trials {' '.join(f'{t:.4f}' for t in trials)}", flush=True)
