#!/usr/bin/env python3
"""Time the coarse part of a V-cycle (N=64 hierarchy) with and without the tail kernel."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multigrid_poisson_solver_amd as mg
mg.init(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
tmp = tempfile.mkdtemp()
path = os.path.join(tmp, "v.txt")
mg.write_vcycle_file(path, N, 8, 3, 1e-7)
plan = mg.CyclePlan(path, fused=True, report=False, error=False)
for _ in range(5):
    plan.execute()
mg.sync()
t = []
for _ in range(50):
    r = plan.execute()
    t.append(r["device_ms"])
t.sort()
print(os.environ.get("TAG", ""), f"N={N} V-cycle device ms: median {t[len(t)//2]*1e3:.1f} us  min {t[0]*1e3:.1f} us")
