import os, sys, tempfile
sys.path.insert(0, "/root/repo")
import multigrid_poisson_solver_amd as mg
mg.init(0)
tmp = tempfile.mkdtemp()
for kind, N in (("V", 64), ("W", 64)):
    path = os.path.join(tmp, f"{kind}.txt")
    (mg.write_vcycle_file if kind == "V" else mg.write_wcycle_file)(path, N, 8, 3, 1e-7)
    plan = mg.CyclePlan(path, fused=True, report=False, error=False)
    r = plan.execute()
    print(kind, "GS iterations of last solve:", mg.lastExactSolverIterations(), flush=True)
    plan.close()
