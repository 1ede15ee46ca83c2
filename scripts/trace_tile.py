#!/usr/bin/env python3
"""In-kernel timeline of the register-tile node kernels (a -DMG_TILE_TRACE build selected with MG_LIB): one V-cycle
from N (default 1024), the trace lines of its last run on stderr."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multigrid_poisson_solver_amd as mg

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
mg.init(0)
path = os.path.join(tempfile.mkdtemp(), "v.txt")
mg.write_vcycle_file(path, N, 8, 3, 1e-7)
plan = mg.CyclePlan(path, fused=True, report=False, error=False)
for i in range(3):
    print(f"--- run {i}", file=sys.stderr, flush=True)
    plan.execute()
