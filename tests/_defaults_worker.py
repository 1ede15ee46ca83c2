"""Child process of test_cycle_gpu.py::test_product_thresholds_*: one V(3,3)-cycle per size with the LIBRARY'S OWN
thresholds (the test suite lowers MG_RECOMPUTE_MIN_N / MG_NT_MIN_N / MG_F32_COLS4_MIN_N so that small grids run the
large-grid code paths; the caller removed those variables), final U as a 128-bit device checksum."""
import ctypes as C
import json
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multigrid_poisson_solver_amd as mg

mg.init(0)
out = {}
for N in (int(a) for a in sys.argv[1:]):
    path = os.path.join(tempfile.mkdtemp(), f"V{N}.txt")
    mg.write_vcycle_file(path, N, 8, 3, 1e-7)
    plan = mg.CyclePlan(path, fused=True, report=False)
    r = plan.execute()
    s = (C.c_uint64 * 2)()
    mg.lib().mg_checksum(r["U_ptr"], N * N, s)
    out[str(N)] = {"status": r["status"], "sum": [int(s[0]), int(s[1])], "mg_error": r["mg_error"],
                   "errors": [rec[3] for rec in r["records"]]}
    plan.close()
    mg.lib().mg_pool_trim()
print("DEFAULTS_WORKER " + json.dumps(out), flush=True)
