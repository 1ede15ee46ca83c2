"""W-cycle on a hierarchy that is not a power of two, batched schedule (or its fallback) against the oracle:
   python scripts/check_w_odd.py N N_min steps     (one case per process; run under `timeout`)"""
import os, sys
os.environ.setdefault("OMP_NUM_THREADS", "16")      # (the oracle: a team of every hardware thread of the box spins for minutes)
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import multigrid_poisson_solver_amd as mg
import _oracle
N, nmin, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mg.init(0); orc = _oracle.Oracle()
path = "/tmp/W%d.txt" % N
mg.write_wcycle_file(path, N, nmin, steps, 1e-7)
want = orc.run_cycle_file(path, want_report=False)
print(N, "oracle done, gs iterations", orc.gs_iterations(), flush=True)
plan = mg.CyclePlan(path, fused=True, report=False)
got = plan.execute(fetch_U=True)
ok = got["status"] == 0 and np.array_equal((got["U"] + 0.0).view(np.uint64), (want["U"] + 0.0).view(np.uint64))
recs = len(got["records"]) == len(want["records"]) and all(tuple(g[:3]) == tuple(w[:3]) and abs(g[3] - w[3]) <= 1e-12 * abs(w[3]) + 1e-300 for g, w in zip(got["records"], want["records"]))
print(N, "status", got["status"], want["status"], "launches", got["schedule_launches"], "U bits", ok, "records", recs, "gs", mg.lastExactSolverIterations(), orc.gs_iterations(), "device ms %.3f" % got["device_ms"], flush=True)
plan.close()
