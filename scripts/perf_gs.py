#!/usr/bin/env python3
"""Back-to-back coarse solves and small V-cycles: per-call device time."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import multigrid_poisson_solver_amd as mg
mg.init(0)
for N in (8, 16, 32):
    F = mg.DeviceGrid.from_host(np.random.default_rng(1).random((N, N)) - 0.5)
    U = mg.DeviceGrid(N)
    for _ in range(5):
        mg.doExactSolver(N, 1.0, U, F, 1e-7, 1)
    it = mg.lastExactSolverIterations()
    mg.sync()
    t = time.perf_counter()
    for _ in range(200):
        mg.doExactSolver(N, 1.0, U, F, 1e-7, 1)
    mg.sync()
    dt = (time.perf_counter() - t) / 200
    print(f"GS N={N}: {dt*1e6:.1f} us per solve, {it} iterations, {dt*1e9/it:.0f} ns per iteration")
# the tail alone: V-cycle whose finest level is 64
tmp = tempfile.mkdtemp()
path = os.path.join(tmp, "v.txt")
mg.write_vcycle_file(path, 64, 8, 3, 1e-7)
plan = mg.CyclePlan(path, fused=True, report=False, error=False)
for _ in range(5):
    plan.execute()
t = sorted(plan.execute()["device_ms"] for _ in range(50))
print(f"V-cycle 64..8 (1 smoothing launch pair at 64 + tail 32..8): median {t[25]*1e3:.1f} us")
