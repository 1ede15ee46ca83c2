"""In-kernel timeline (MG_TAIL_TRACE=1) of a coarse-tail launch that starts at N = 64: the W-cycle's unit of work."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multigrid_poisson_solver_amd as mg
mg.init(0)
path = os.path.join(tempfile.mkdtemp(), "W.txt")
mg.write_wcycle_file(path, 128, 8, 3, 1e-7)
plan = mg.CyclePlan(path, fused=True, report=False, error=False)
plan.execute()
plan.close()
