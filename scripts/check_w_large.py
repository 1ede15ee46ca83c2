"""W(3,3)-cycle at a large size as a batched schedule against the oracle's own run: checksum of the final U, every record.\n   python scripts/check_w_large.py [N]   (N = 16384: 19 launches, ~4.2 ms; the oracle ~10 s on 16 threads)"""
import os, sys, time, ctypes as C
os.environ.setdefault("OMP_NUM_THREADS", "16")
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import multigrid_poisson_solver_amd as mg
import _oracle, _synth
mg.init(0)
orc = _oracle.Oracle(); orc.set_threads(16)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
path = "/tmp/W%d.txt" % N
mg.write_wcycle_file(path, N, 8, 3, 1e-7)
t = time.time(); want = orc.run_cycle_file(path, want_report=False); print("oracle %.1f s" % (time.time() - t), flush=True)
plan = mg.CyclePlan(path, fused=True, report=False)
got = plan.execute()
s = (C.c_uint64 * 2)()
mg.lib().mg_checksum(got["U_ptr"], N * N, s)
ws = _synth.checksum(want["U"])
print("status", got["status"], want["status"], "launches", got["schedule_launches"], "device ms", got["device_ms"])
print("checksum equal:", (int(s[0]), int(s[1])) == tuple(ws))
bad = [i for i, (g, w) in enumerate(zip(got["records"], want["records"])) if tuple(g[:3]) != tuple(w[:3]) or abs(g[3] - w[3]) > 1e-12 * abs(w[3]) + 1e-300]
print("records", len(got["records"]), len(want["records"]), "mismatches", len(bad), "mg_error", got["mg_error"], want["mg_error"])
for _ in range(3): plan.enqueue()
r = plan.collect(); print("window ms", r["device_ms"])
