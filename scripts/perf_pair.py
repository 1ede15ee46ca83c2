import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multigrid_poisson_solver_amd as mg
mg.init(0)
for N in (8192, 16384, 4096):
    Ua, Ub, Ff = mg.DeviceGrid.uniform(N, 3), mg.DeviceGrid(N), mg.DeviceGrid.uniform(N, 7)
    for _ in range(5): mg.smooth_pp(N, 1.0, Ua, Ub, Ff, 1)
    mg.sync(); mg.profile_begin(min_N=N)
    for _ in range(20): mg.smooth_pp(N, 1.0, Ua, Ub, Ff, 1)
    e = mg.profile_end()[0]; avg = e["total_ms"] / e["launches"]
    print(os.environ.get("MG_LIB", "product")[-12:], N, e["name"], "%.1f us %.0f GB/s frac %.3f" % (avg * 1e3, 24.0 * N * N / avg / 1e6, 24.0 * N * N / avg / 1e6 / 8000))
    for g in (Ua, Ub, Ff): g.free()
