#!/usr/bin/env python3
"""Time the two fused node kernels at one size (live hipEvent pairs): a quick A/B harness
for kernel tuning knobs set through environment variables."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multigrid_poisson_solver_amd as mg

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = 20
mg.init(0)
M = N // 2
F = mg.DeviceGrid.uniform(N, 7)
U1, U2, Fc, Uc = mg.DeviceGrid.uniform(N, 3), mg.DeviceGrid(N), mg.DeviceGrid(M), mg.DeviceGrid.uniform(M, 8)
for _ in range(3):
    mg.smooth_restrict(N, 1.0, None, U2, F, 3, M, Fc)
    mg.prolong_smooth(M, Uc, N, 1.0, U1, U2, F, 3)
mg.sync()
mg.profile_begin(0)
for _ in range(reps):
    mg.smooth_restrict(N, 1.0, None, U2, F, 3, M, Fc)
    mg.prolong_smooth(M, Uc, N, 1.0, U1, U2, F, 3)
    mg.smooth_pp(N, 1.0, U1, U2, F, 3)
    mg.smooth_pp(N, 1.0, None, U2, F, 3)
for e in mg.profile_end():
    avg = e["total_ms"] / e["launches"]
    print(f"{os.environ.get('TAG', ''):>16} N={N} {e['name']:<36} {avg * 1e3:8.1f} us  algo {e['algo_bytes'] / avg / 1e6:8.0f} GB/s")
