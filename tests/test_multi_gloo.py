"""CPU coverage of the N > 1 path: world_size-2 (and 3) gloo runs of tests/_slab_worker.py.
The worker runs in its own processes (it imports torch before the engine library; this
pytest process never imports torch)."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world,N,collapse,ca_mode,ca_pct", [(2, 256, 32, 1, 10), (2, 128, 64, 0, 10), (3, 256, 64, 1, 100),
                                                             (2, 256, 32, 2, 100), (2, 512, 32, 1, 100),
                                                             # the target rank count of BASELINE configs[3]/[4]: 8 processes (a GPU box
                                                             # admits at most 6 processes on its card, so 8 RANKS can only run here)
                                                             (8, 1024, 64, 1, 10)])
def test_row_slab_schedule_on_gloo(tmp_path, world, N, collapse, ca_mode, ca_pct):
    """ca_mode 0: every halo exchanged (one group per level); 1: F halos recomputed while the extra rows stay below
    ca_pct per cent of a slab (100: always); 2: U halos recomputed as well (no ghost exchange, only the all-gather)."""
    import multigrid_poisson_solver_amd as mg
    path = str(tmp_path / "V.txt")
    mg.write_vcycle_file(path, N, 8, 3, 1e-7)
    env = dict(os.environ, OMP_NUM_THREADS="2", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(HERE, "_slab_worker.py"), str(N), str(collapse), "3", path, str(ca_mode), str(ca_pct)]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert f"SLAB_EMULATION OK {world} {N} {collapse} mode {ca_mode}" in out.stdout
    if ca_mode == 2:
        assert "exchanges 0" in out.stdout


def test_partition_properties():
    """mg_slab_partition (host code of the product): contiguous cover of every distributed
    level, every slab at least two halos tall, coarse rows owned where their lower-left
    restriction sample lives."""
    import multigrid_poisson_solver_amd as mg
    G = mg.slab_ghost_rows()
    assert G >= 6
    # the bench configurations WITH THE PRODUCT'S THRESHOLDS (this process runs with the test suite's lower ones, see
    # conftest.py: a child process): the halos of the schedule fit, nothing but the last level's U halo travels, and
    # the levels whose slabs are large enough recompute their pre-smoothed field (such a level has no U halo at all)
    code = """
import multigrid_poisson_solver_amd as mg
for N, R in [(16384, 8), (23040, 8), (11520, 2), (16384, 4), (16384, 2)]:
    sched = [d for d in mg.slab_schedule(N, 8, R, 1024, 3) if not d["collapsed"]]
    assert all(d["xF"] == 0 for d in sched) and [d["xU"] > 0 for d in sched] == [False] * (len(sched) - 1) + [True], (N, R)
    # (the recomputing pair where a LAUNCH has at least 4096^2 / 2 points: N * rows per slab)
    assert [d["pre"] for d in sched] == [3 if d["N"] * (d["N"] // R) >= 4096 * 4096 // 2 else 0 for d in sched], (N, R)
    assert all(d["xU"] == 0 for d in sched if d["pre"])
    for d in sched:
        rows = min(hi - lo for lo, hi in d["own"])
        assert d["halo"] <= rows // 8, (N, R, d["N"], d["halo"], rows)
        for r in range(R):
            assert d["dext"][r][0] <= d["own"][r][0] and d["dext"][r][1] >= d["own"][r][1]
print("SCHEDULE_OK")
"""
    env = {k: v for k, v in os.environ.items() if k not in ("MG_RECOMPUTE_MIN_N", "MG_NT_MIN_N")}
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert "SCHEDULE_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
    # with the thresholds of the tests: every recomputing level is free of U halos, slabs contain their dext rows
    for N, R in [(16384, 8), (23040, 8), (11520, 2)]:
        sched = [d for d in mg.slab_schedule(N, 8, R, 1024, 3) if not d["collapsed"]]
        assert all(d["xU"] == 0 for d in sched if d["pre"])
        for d in sched:
            for r in range(R):
                assert d["dext"][r][0] <= d["own"][r][0] and d["dext"][r][1] >= d["own"][r][1]
    for N, R, collapse in [(8192, 8, 1024), (23040, 8, 1024), (11520, 2, 1024), (23168, 8, 1024), (1024, 3, 128), (16384, 4, 512)]:
        levels = mg.slab_partition(N, 8, R, collapse)
        seen_collapsed = False
        for li, (n, collapsed, ranges) in enumerate(levels):
            if collapsed:
                seen_collapsed = True
                assert ranges[0] == (0, n) and all(r == (0, 0) for r in ranges[1:])
                continue
            assert not seen_collapsed and n % 2 == 0 and n > collapse
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            for a, b in zip(ranges[:-1], ranges[1:]):
                assert a[1] == b[0]
            assert min(hi - lo for lo, hi in ranges) >= 2 * G
            if li > 0 and not levels[li - 1][1]:
                fine = levels[li - 1][2]
                lo_t, _ = mg.restriction_table(levels[li - 1][0], n)
                for r, (clo, chi) in enumerate(ranges):
                    for rc in (max(1, clo), min(n - 2, chi - 1)):
                        assert fine[r][0] <= lo_t[rc] < fine[r][1]
