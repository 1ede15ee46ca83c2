"""The 1-D row-slab decomposition, verified bit for bit on ONE GPU with virtual ranks: all R
slabs live in this process and exchange ghost rows by device copies (mg_slab_load rank=-1).
The same driver code runs one rank per process over RCCL under bench.py --gpus N; only the
transport (mg_comm.cpp) differs."""
import os

import numpy as np
import pytest

from conftest import assert_bits

pytestmark = pytest.mark.gpu


def check(got, U, want):
    assert got["status"] == 0 and want["status"] == 0
    assert_bits(U, want["U"], "final U", zero_sign=True)
    assert got["mg_error"] == pytest.approx(want["mg_error"], rel=1e-10)
    assert len(got["records"]) == len(want["records"])
    for g, w in zip(got["records"], want["records"]):
        assert tuple(g[:3]) == tuple(w[:3])
        assert g[3] == pytest.approx(w[3], rel=1e-12, abs=1e-300)


@pytest.mark.parametrize("R", [1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("collapse", [64, 256])
def test_virtual_slabs_vcycle_vs_oracle(mg, oracle, tmp_path, R, collapse):
    N = 1024
    path = str(tmp_path / "V.txt")
    mg.write_vcycle_file(path, N, 8, 3, 1e-7)
    want = oracle.run_cycle_file(path)
    plan = mg.SlabPlan(path, R, -1, collapse)
    for _ in range(2):  # the second run reuses pooled buffers
        got = plan.execute()
        check(got, plan.gather_U(N), want)
    plan.close()


@pytest.mark.parametrize("R", [2, 5])
def test_virtual_slabs_wcycle_and_other_steps(mg, oracle, tmp_path, R):
    N = 512
    w = str(tmp_path / "W.txt")
    mg.write_wcycle_file(w, N, 8, 2, 1e-7)
    want = oracle.run_cycle_file(w)
    plan = mg.SlabPlan(w, R, -1, 64)
    got = plan.execute()
    check(got, plan.gather_U(N), want)
    plan.close()
    v = str(tmp_path / "V4.txt")
    mg.write_vcycle_file(v, N, 16, 4, 1e-6)
    want = oracle.run_cycle_file(v)
    plan = mg.SlabPlan(v, R, -1, 128)
    got = plan.execute()
    check(got, plan.gather_U(N), want)
    plan.close()


@pytest.mark.parametrize("R,N,levels", [(2, 1448, 8), (8, 1448, 8), (2, 1440, 8), (8, 1440, 8)])
def test_virtual_slabs_on_the_weak_scaling_size_family(mg, oracle, tmp_path, R, N, levels):
    """bench.py --gpus 2/8 runs N = 11520 / 23040 = 2^k * 45: every level above the coarse tail is even,
    the tail runs 45 -> 22 -> 11 (one-wave LDS Gauss-Seidel on 11 x 11).  Same chain at 1/8 of the size
    (1440), and the harder family 2^k * 181 (1448 -> 724 distributed, 362 -> 181 -> 90 -> 45 -> 22 -> 11
    collapsed: an odd level above the tail, non-nested transfers)."""
    path = str(tmp_path / f"V{N}.txt")
    assert mg.write_vcycle_file(path, N, 8, 3, 1e-7) == levels
    want = oracle.run_cycle_file(path)
    plan = mg.SlabPlan(path, R, -1, 400)
    got = plan.execute()
    check(got, plan.gather_U(N), want)
    plan.close()
    single = mg.CyclePlan(path, fused=True, report=False)
    ref = single.execute(fetch_U=True)
    assert_bits(ref["U"], want["U"], "single-GPU driver on the same file", zero_sign=True)
    single.close()


@pytest.mark.parametrize("N,mixed,refine", [(4096, False, 1), (8192, False, 1), (8192, True, 1), (16384, False, 1), (32768, True, 1),
                                            (32768, True, 2)])
def test_virtual_slabs_match_single_gpu_driver_at_full_size(mg, tmp_path, N, mixed, refine):
    """8 slabs of a V-cycle (fp64 and fp32 fields) against the single-GPU driver (itself pinned to the
    oracle up to 8192^2, tests/test_cycle_gpu.py).  16384^2 fp64 and 32768^2 mixed are the sizes of
    BASELINE.json configs[3] and configs[4]; there the results are compared through the checksum.
    (32768, mixed, refine = 2) is configs[4] AS WRITTEN -- "mixed fp32 smoothing / fp64 residual correction": two fp32
    cycles joined by the fp64 residual of the fp64 iterate and an fp64 correction, on 8 slabs and on one GPU: the fp64
    iterate (checksum), every smoothing error and the residual norm of the refinement must agree."""
    path = str(tmp_path / f"V{N}.txt")
    mg.write_vcycle_file(path, N, 8, 3, 1e-7)
    single = mg.CyclePlan(path, fused=True, report=False, mixed=mixed, refinement=refine)
    big = N > 8192
    ref = single.execute(fetch_U=not big)
    plan = mg.SlabPlan(path, 8, -1, 1024 if N >= 8192 else 512, mixed=mixed, refinement=refine)
    got = plan.execute()
    assert got["status"] == 0 and ref["status"] == 0
    if refine > 1:
        assert len(got["refinement_errors"]) == refine - 1 and len(ref["refinement_errors"]) == refine - 1
        for g, w in zip(got["refinement_errors"], ref["refinement_errors"]):
            assert g == pytest.approx(w, rel=1e-12) and g > 0
    if big:
        import ctypes as C
        import _synth
        out = (C.c_uint64 * 2)()
        mg.lib().mg_checksum(ref["U_ptr"], N * N, out)
        assert _synth.checksum(plan.gather_U(N)) == (int(out[0]), int(out[1])), "8 slabs vs 1 GPU"
    else:
        assert_bits(plan.gather_U(N), ref["U"], "8 slabs vs 1 GPU", zero_sign=True)
    assert got["mg_error"] == pytest.approx(ref["mg_error"], rel=1e-10)
    for g, w in zip(got["records"], ref["records"]):
        assert tuple(g[:3]) == tuple(w[:3]) and g[3] == pytest.approx(w[3], rel=1e-12, abs=1e-300)
    plan.close(); single.close()


def test_config4_size_vs_oracle(mg, oracle, tmp_path):
    """BASELINE.json configs[3] at full size against the ORACLE itself (not only against the single-GPU driver): the
    V(3,3)-cycle at N = 16384^2, single-GPU fused driver and 8 row slabs (communication-avoiding schedule, NaN-poisoned
    slab arrays), final U through the 128-bit checksum, every smoothing error, the analytic error."""
    import ctypes as C
    import _synth
    N = 16384
    path = str(tmp_path / f"V{N}.txt")
    mg.write_vcycle_file(path, N, 8, 3, 1e-7)
    want = oracle.run_cycle_file(path, want_report=False)
    assert want["status"] == 0
    want_sum = _synth.checksum(want["U"])
    want_err, want_rec = want["mg_error"], want["records"]
    del want
    single = mg.CyclePlan(path, fused=True, report=False)
    got = single.execute()
    out = (C.c_uint64 * 2)()
    mg.lib().mg_checksum(got["U_ptr"], N * N, out)
    assert (int(out[0]), int(out[1])) == want_sum, "single-GPU driver at 16384^2 differs from the oracle"
    assert got["mg_error"] == pytest.approx(want_err, rel=1e-10)
    single.close()
    mg.lib().mg_pool_trim()
    plan = mg.SlabPlan(path, 8, -1, 1024)
    res = plan.execute()
    assert res["status"] == 0
    assert _synth.checksum(plan.gather_U(N)) == want_sum, "8 row slabs at 16384^2 differ from the oracle"
    assert res["mg_error"] == pytest.approx(want_err, rel=1e-10)
    for g, w in zip(res["records"], want_rec):
        assert tuple(g[:3]) == tuple(w[:3]) and g[3] == pytest.approx(w[3], rel=1e-12, abs=1e-300)
    plan.close()
    mg.lib().mg_pool_trim()
    # ... and the same size with the LIBRARY'S OWN thresholds (a child process without this suite's MG_* overrides: the
    # configuration bench.py's strong-scaling base runs) against the oracle as well (it used to be compared
    # with this process's run only)
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("MG_RECOMPUTE_MIN_N", "MG_NT_MIN_N", "MG_F32_COLS4_MIN_N")}
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "_defaults_worker.py"), str(N)], env=env,
                         capture_output=True, text=True, timeout=600)
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("DEFAULTS_WORKER ")]
    assert out.returncode == 0 and line, out.stdout[-2000:] + out.stderr[-3000:]
    child = json.loads(line[0][len("DEFAULTS_WORKER "):])[str(N)]
    assert child["status"] == 0 and tuple(child["sum"]) == want_sum, "product thresholds at 16384^2 differ from the oracle"
    assert child["mg_error"] == pytest.approx(want_err, rel=1e-10)
    for g, w in zip(child["errors"], want_rec):
        assert g == pytest.approx(w[3], rel=1e-12, abs=1e-300)


def test_slab_mode_refuses_what_it_does_not_implement(mg, tmp_path):
    trig = tmp_path / "trigger.txt"
    trig.write_text("1.0 0.0 0.0\n-1 1\n256 8\n-1\n-1\n0\n0.0000001 1\n1\n1\n2")
    with pytest.raises(mg.MGError, match="row-slab mode"):
        mg.SlabPlan(str(trig), 2, -1, 64)
    small = tmp_path / "small.txt"
    small.write_text("1.0 0.0 0.0\n3 1\n32 8\n-1\n0\n0.0000001 1\n1\n2")
    with pytest.raises(mg.MGError, match="too small"):
        mg.SlabPlan(str(small), 8, -1, 8)


def test_rccl_communicator_single_rank(mg):
    """The transport itself cannot be exercised across ranks on a one-GPU box (RCCL refuses
    two ranks on one device); a 1-rank communicator at least proves the library binding."""
    uid = mg.comm_unique_id()
    assert len(uid) == mg.lib().mg_comm_unique_id_bytes()
    mg.comm_init(0, 1, uid)
    assert mg.lib().mg_comm_rank() == 0 and mg.lib().mg_comm_size() == 1
    # the calls the slab driver makes, through the lazily resolved entry points: a grouped ncclSend/ncclRecv pair
    # (to this rank itself) on a second stream ordered by events, then ncclAllGather; the bytes must arrive
    for n in (1, 4096, 1 << 20):
        assert mg.lib().mg_comm_selftest(n) == 0
        assert mg.lib().mg_last_error() == 0, mg.lib().mg_last_error_string()
    mg.lib().mg_comm_finalize()


# (world = 4 is the most a GPU box admits: at most 6 processes may have its card open at once -- the ranks, this pytest
# process and the launcher; a world of 5 was killed by the box's process guard.  The 8-rank case runs as 8 virtual ranks on
# the GPU, as 8 gloo processes on the CPU -- tests/test_multi_gloo.py -- and on the driver's 8-GPU node)
@pytest.mark.parametrize("world,N,collapse,mixed", [(2, 512, 64, False), (3, 1024, 128, False), (4, 1024, 256, False),
                                                    (3, 1024, 128, True), (2, 1024, 128, 3)])
def test_rank_mode_over_host_transport(mg, oracle, tmp_path, world, N, collapse, mixed):
    """The driver in RANK mode (one process per slab, as under RCCL) with the host-staged
    transport over gloo: `world` processes on this one GPU, every rank's owned rows bit-identical
    to the oracle's cycle, errors combined over ranks (tests/_slab_host_worker.py)."""
    import socket
    import subprocess
    import sys
    path = str(tmp_path / "V.txt")
    mg.write_vcycle_file(path, N, 8, 3, 1e-7)
    if mixed:   # fp32 slabs: the single-GPU fp32 cycle (itself pinned to the numpy restatement) is the yardstick
        one = mg.CyclePlan(path, fused=True, mixed=True, refinement=int(mixed))   # True -> 1 cycle, 3 -> refinement
        want = one.execute(fetch_U=True)
        one.close()
    else:
        want = oracle.run_cycle_file(path)
    assert want["status"] == 0
    want_path = str(tmp_path / "want.npz")
    np.savez(want_path, N=N, U=want["U"], mg_error=want["mg_error"],
             rec_nodes=np.array([(r[0], r[1]) for r in want["records"]]),
             rec_errors=np.array([r[3] for r in want["records"]]))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, OMP_NUM_THREADS="2", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(here, "_slab_host_worker.py"), path, want_path, str(collapse)] + ([f"mixed{int(mixed)}"] if mixed else [])
    out = subprocess.run(cmd, cwd=os.path.dirname(here), env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert f"SLAB_HOST_TRANSPORT OK {world} {N} {collapse}" in out.stdout


def test_bench_self_launches_its_ranks(mg):
    """`python3 bench.py --gpus 2` outside torchrun must start its two ranks itself (a child torch.distributed.run,
    before the parent touches the GPU) and relay rank 0's ONE JSON line: the shape of the command the driver uses
    for the scaling run.  Two RCCL ranks cannot share this box's one GPU, so the wire is the host-staged
    rehearsal transport; the line must carry every leg (weak = `value`, strong = N 16384^2, the headline grid 8192^2 cut
    into slabs)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MG_SLAB_POISON")}
    env.update(MG_BENCH_TRANSPORT="host", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-large"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["metric"] == "vcycle_mlups" and line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "weak"
    assert line["config"]["N"] == 11520 and line["value"] > 0
    assert line["strong_scaling"]["N"] == 16384 and line["strong_scaling"]["value"] > 0
    # BASELINE.json's metric reads "8192^2 fp64, 1/2/4/8 GPUs": the headline grid itself cut into slabs is in every line
    assert line["strong_scaling_8192"]["N"] == 8192 and line["strong_scaling_8192"]["n_gpus"] == 2 and line["strong_scaling_8192"]["value"] > 0
    assert line["strong_scaling_8192"]["mg_error"] == pytest.approx(0.000883, abs=2e-6)
    assert 0 < line["mg_error"] < 1e-3                                     # 11520 -> ... -> 11: another hierarchy, another error
    assert line["strong_scaling"]["mg_error"] == pytest.approx(0.000883, abs=2e-6)   # the V(3,3) result of the 2^k hierarchies
    # the line certifies its own wire: the communicator's rank count (not the environment's), where the transport came from,
    # every rank's own clock
    assert line["rccl"]["nranks"] == 2 and line["rccl"]["ranks_agree"] is True and "host" in line["rccl"]["transport"]
    assert len(line["ms_per_step_ranks"]["all"]) == 2 and line["ms_per_step_ranks"]["max"] == pytest.approx(line["ms_per_step"], rel=1e-3)


@pytest.mark.parametrize("ca_mode,ca_pct", [(0, 10), (1, 100), (2, 100), (1, 10)])
@pytest.mark.parametrize("R,N,collapse", [(8, 2048, 64), (3, 1024, 128), (5, 4096, 256)])
def test_virtual_slabs_every_schedule(mg, oracle, tmp_path, monkeypatch, R, N, collapse, ca_mode, ca_pct):
    """The three schedules of mg_slab.cpp -- every halo exchanged (0), F halos recomputed (1, here with and without
    the 10 % cap), U halos recomputed too (2) -- give the oracle's bits; fresh slab arrays are NaN (MG_SLAB_POISON), so
    a halo row that was neither recomputed nor exchanged would show."""
    monkeypatch.setenv("MG_SLAB_CA", str(ca_mode))
    monkeypatch.setenv("MG_SLAB_CA_PCT", str(ca_pct))
    path = str(tmp_path / "V.txt")
    mg.write_vcycle_file(path, N, 8, 3, 1e-7)
    want = oracle.run_cycle_file(path)
    plan = mg.SlabPlan(path, R, -1, collapse)
    for _ in range(2):
        got = plan.execute()
        check(got, plan.gather_U(N), want)
    for _ in range(3):   # back-to-back windows: exchanges of consecutive windows must not overtake each other
        plan.enqueue()
    got = plan.collect()
    check(got, plan.gather_U(N), want)
    plan.close()


def test_fresh_slab_arrays_are_poisoned(mg, tmp_path):
    """conftest sets MG_SLAB_POISON: the rows of a slab array that nobody has written are NaN, so a
    missing halo row cannot hide behind stale data in the parity tests above."""
    path = str(tmp_path / "V.txt")
    mg.write_vcycle_file(path, 512, 8, 3, 1e-7)
    plan = mg.SlabPlan(path, 2, -1, 64)
    assert np.isnan(plan.gather_U(512)).all()
    assert plan.execute()["status"] == 0
    assert not np.isnan(plan.gather_U(512)).any()
    plan.close()


@pytest.mark.parametrize("steps", [1, 2, 4])
def test_virtual_slabs_deep_hierarchy_other_sweep_counts(mg, oracle, tmp_path, steps):
    """8 slabs, five distributed levels, 1 / 2 / 4 sweeps per node: the redundant halo rows of the ascent
    (up to 8 per side at 4 sweeps) plus the rows a launch reads beyond them have to fit the halo."""
    N = 2048
    path = str(tmp_path / "V.txt")
    mg.write_vcycle_file(path, N, 8, steps, 1e-7)
    sched = [d for d in mg.slab_schedule(N, 8, 8, 64, steps) if not d["collapsed"]]
    assert len(sched) == 4 and all(d["halo"] <= min(hi - lo for lo, hi in d["own"]) for d in sched)
    want = oracle.run_cycle_file(path)
    plan = mg.SlabPlan(path, 8, -1, 64)
    got = plan.execute()
    check(got, plan.gather_U(N), want)
    plan.close()
