"""Child process of test_cycle_gpu.py::test_product_thresholds_*: one cycle per spec with the LIBRARY'S OWN thresholds
(the test suite lowers MG_RECOMPUTE_MIN_N / MG_NT_MIN_N / MG_F32_COLS4_MIN_N so that small grids run the large-grid
code paths; the caller removed those variables), final U as a 128-bit device checksum.
A spec is `N` (= V:N:3) or `KIND:N:STEPS` with KIND in V, W (fp64 cycles), MV (the fp32 cycle of the mixed mode)."""
import ctypes as C
import json
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multigrid_poisson_solver_amd as mg

mg.init(0)
out = {}
for spec in sys.argv[1:]:
    kind, N, steps = (("V", spec, "3") if ":" not in spec else spec.split(":"))
    N, steps = int(N), int(steps)
    path = os.path.join(tempfile.mkdtemp(), f"{kind}{N}.txt")
    (mg.write_wcycle_file if kind == "W" else mg.write_vcycle_file)(path, N, 8, steps, 1e-7)
    plan = mg.CyclePlan(path, fused=True, report=False, mixed=(kind == "MV"))
    r = plan.execute()
    s = (C.c_uint64 * 2)()
    mg.lib().mg_checksum(r["U_ptr"], N * N, s)
    out[spec] = {"status": r["status"], "sum": [int(s[0]), int(s[1])], "mg_error": r["mg_error"],
                 "errors": [rec[3] for rec in r["records"]]}
    plan.close()
    mg.lib().mg_pool_trim()
print("DEFAULTS_WORKER " + json.dumps(out), flush=True)
