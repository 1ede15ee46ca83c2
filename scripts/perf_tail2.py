#!/usr/bin/env python3
"""Where the coarse-tail kernel's time goes: V-cycles whose finest level is already a tail level,
for several depths and sweep counts (device time between the window's two events, median)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multigrid_poisson_solver_amd as mg
mg.init(0)
tmp = tempfile.mkdtemp()


def run(top, n_min, steps, tol, kind="V"):
    path = os.path.join(tmp, f"{kind}_{top}_{n_min}_{steps}.txt")
    (mg.write_vcycle_file if kind == "V" else mg.write_wcycle_file)(path, top, n_min, steps, tol)
    plan = mg.CyclePlan(path, fused=True, report=False, error=False)
    for _ in range(5):
        plan.execute()
    t = sorted(plan.execute()["device_ms"] for _ in range(40))
    plan.close()
    return t[len(t) // 2] * 1e3


for top, n_min, steps, tol, kind in [(8, 8, 3, 1e-7, "V"), (8, 8, 3, 1e-1, "V"), (16, 8, 3, 1e-1, "V"), (16, 8, 1, 1e-1, "V"),
                                     (32, 8, 3, 1e-1, "V"), (64, 8, 3, 1e-1, "V"), (64, 8, 1, 1e-1, "V"), (64, 32, 3, 1e-1, "V"),
                                     (64, 32, 1, 1e-1, "V"), (64, 8, 3, 1e-7, "V"), (64, 8, 3, 1e-7, "W"), (64, 8, 3, 1e-1, "W")]:
    print(f"{kind} top={top:3d} n_min={n_min:3d} steps={steps} tol={tol:g}: {run(top, n_min, steps, tol, kind):8.1f} us", flush=True)
