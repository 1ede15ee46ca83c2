"""The mixed-precision mode: fp32 fields and arithmetic for the whole
cycle, fp64 source rounded once, result widened to fp64.  The HIP kernels are held bit for bit to
the numpy fp32 restatement (tests/_oracle_f32.py, "parity unpinned": the reference's fp32 path is
CUDA-only) and, against the fp64 oracle, to a stated tolerance."""
import numpy as np
import pytest

import _oracle_f32 as o32

pytestmark = pytest.mark.gpu


def bits32(a, b):
    a, b = np.ascontiguousarray(a, dtype=np.float32), np.ascontiguousarray(b, dtype=np.float32)
    return a.shape == b.shape and np.array_equal((a + np.float32(0)).view(np.uint32), (b + np.float32(0)).view(np.uint32))


@pytest.mark.parametrize("N", [8, 16, 64, 256, 1024, 1448, 2048])
@pytest.mark.parametrize("step", [1, 3, 4])
def test_fused_nodes_fp32_vs_numpy(mg, N, step):
    M = N // 2
    rng = np.random.default_rng(N + step)
    F = (rng.random((N, N)) - 0.5).astype(np.float32)
    Fd, Uo, Fc = mg.DeviceGrid32.from_host(F), mg.DeviceGrid32((N, N)), mg.DeviceGrid32((M, M))
    err = mg.smooth_restrict_f32(N, 1.0, Uo, Fd, step, M, Fc, want_error=True)
    U, e = o32.smooth(np.zeros((N, N), dtype=np.float32), F, step, 1.0)
    assert bits32(Uo.to_host(), U), f"fp32 smoothing N={N} step={step}"
    assert err == pytest.approx(e, rel=1e-12)
    assert bits32(Fc.to_host(), o32.restrict_neg_residual(mg, U, F, 1.0, M)), "fp32 fused restriction"
    if M >= 4:
        Uc = (rng.random((M, M)) - 0.5).astype(np.float32)
        Uf = rng.random((N, N)).astype(np.float32)
        out = mg.DeviceGrid32((N, N))
        err = mg.prolong_smooth_f32(M, mg.DeviceGrid32.from_host(Uc), N, 1.0, mg.DeviceGrid32.from_host(Uf), out, Fd, step,
                                    want_error=True)
        want, e = o32.smooth(o32.prolong_add(mg, Uc, Uf), F, step, 1.0)
        assert bits32(out.to_host(), want), f"fp32 prolong+smooth {M}->{N}"
        assert err == pytest.approx(e, rel=1e-12)


@pytest.mark.parametrize("kind,N,n_min", [("V", 256, 8), ("W", 128, 8), ("V", 1024, 8), ("V", 512, 32), ("V", 724, 8), ("W", 362, 8), ("V", 704, 8)])
def test_mixed_cycle_vs_numpy_and_fp64(mg, oracle, tmp_path, kind, N, n_min):
    path = str(tmp_path / "c.txt")
    (mg.write_vcycle_file if kind == "V" else mg.write_wcycle_file)(path, N, n_min, 3, 1e-7)
    toks = open(path).read().split()
    sizes, n = [], N
    while n >= n_min:
        sizes.append(n)
        n //= 2
    F64 = oracle.getSource(N)
    U32, recs = o32.run_cycle_tokens(mg, oracle, F64, 1.0, 3, sizes, toks[7:])
    plan = mg.CyclePlan(path, fused=True, mixed=True)
    got = plan.execute(fetch_U=True)
    assert got["status"] == 0
    assert bits32(got["U"].astype(np.float32), U32), "mixed cycle: fp32 result"
    assert np.array_equal(got["U"], U32.astype(np.float64)), "widening is exact"
    assert len(got["records"]) == len(recs)
    for g, w in zip(got["records"], recs):
        assert (g[0], g[1]) == (w[0], w[1])
        assert g[3] == pytest.approx(w[2], rel=1e-10, abs=1e-300)
    # against the fp64 cycle of the reference: fp32 rounding only (tolerance 2e-5 of max|U|)
    want = oracle.run_cycle_file(path)
    scale = np.abs(want["U"]).max()
    assert np.abs(got["U"] - want["U"]).max() <= 2e-5 * scale
    assert got["mg_error"] == pytest.approx(want["mg_error"], rel=1e-3)
    plan.close()


@pytest.mark.parametrize("steps", [1, 2, 4])
def test_mixed_cycle_other_sweep_counts_vs_numpy(mg, oracle, tmp_path, steps):
    """The fp32 cycle with 1, 2 and 4 sweeps per node (recomputing pair instantiated for 1+1 and 2+2, store/re-read for
    4+4) against the numpy restatement, bit for bit."""
    N, n_min = 1024, 8
    path = str(tmp_path / "c.txt")
    mg.write_vcycle_file(path, N, n_min, steps, 1e-7)
    toks = open(path).read().split()
    sizes, n = [], N
    while n >= n_min:
        sizes.append(n)
        n //= 2
    U32, recs = o32.run_cycle_tokens(mg, oracle, oracle.getSource(N), 1.0, steps, sizes, toks[7:])
    plan = mg.CyclePlan(path, fused=True, mixed=True)
    got = plan.execute(fetch_U=True)
    assert got["status"] == 0
    assert bits32(got["U"].astype(np.float32), U32), f"mixed cycle with {steps} sweeps per node: fp32 result"
    for g, w in zip(got["records"], recs):
        assert (g[0], g[1]) == (w[0], w[1]) and g[3] == pytest.approx(w[2], rel=1e-10, abs=1e-300)
    plan.close()


@pytest.mark.parametrize("N", [181, 90])
def test_fp32_nodes_on_odd_and_non_fusable_sizes(mg, N):
    """Sizes the fused transfer stages do not cover (odd N; the hierarchy 724, 362, 181, 90, 45 of the
    weak-scaling grids) run operator by operator in fp32: same numpy restatement, bit for bit."""
    M = N // 2
    rng = np.random.default_rng(N)
    F = (rng.random((N, N)) - 0.5).astype(np.float32)
    Fd, Uo, Fc = mg.DeviceGrid32.from_host(F), mg.DeviceGrid32((N, N)), mg.DeviceGrid32((M, M))
    err = mg.smooth_restrict_f32(N, 1.0, Uo, Fd, 3, M, Fc, want_error=True)
    U, e = o32.smooth(np.zeros((N, N), dtype=np.float32), F, 3, 1.0)
    assert bits32(Uo.to_host(), U)
    assert err == pytest.approx(e, rel=1e-12)
    assert bits32(Fc.to_host(), o32.restrict_neg_residual(mg, U, F, 1.0, M))
    Uc = (rng.random((M, M)) - 0.5).astype(np.float32)
    Uf = rng.random((N, N)).astype(np.float32)
    out = mg.DeviceGrid32((N, N))
    err = mg.prolong_smooth_f32(M, mg.DeviceGrid32.from_host(Uc), N, 1.0, mg.DeviceGrid32.from_host(Uf), out, Fd, 3, want_error=True)
    want, e = o32.smooth(o32.prolong_add(mg, Uc, Uf), F, 3, 1.0)
    assert bits32(out.to_host(), want)
    assert err == pytest.approx(e, rel=1e-12)


def test_mixed_mode_refuses_unsupported_shapes(mg, tmp_path):
    trig = tmp_path / "t.txt"
    trig.write_text("\n".join(["1.0 0.0 0.0", "-1 1", "64 8", "-1", "-1", "0", "0.0000001 1", "1", "1", "2"]))
    with pytest.raises(mg.MGError, match="mixed-precision"):
        mg.CyclePlan(str(trig), mixed=True)


@pytest.mark.parametrize("N,cycles", [(256, 3), (1024, 2)])
def test_refinement_vs_numpy(mg, oracle, tmp_path, N, cycles):
    """fp32 cycles joined by fp64 residual + fp64 correction: bit for bit against the numpy/oracle
    restatement of the same schedule."""
    path = str(tmp_path / "c.txt")
    mg.write_vcycle_file(path, N, 8, 3, 1e-7)
    toks = open(path).read().split()
    sizes, n = [], N
    while n >= 8:
        sizes.append(n)
        n //= 2
    F64 = oracle.getSource(N)
    want, errs = o32.refine(mg, oracle, F64, 1.0, 3, sizes, toks[7:], cycles)
    plan = mg.CyclePlan(path, fused=True, mixed=True, refinement=cycles)
    for _ in range(2):   # the second window must restore the rounded source first
        got = plan.execute(fetch_U=True)
        assert got["status"] == 0
        assert np.array_equal(got["U"], want), "refined fp64 iterate"
        assert got["refinement_errors"] == pytest.approx(errs, rel=1e-12)
    plan.close()


def test_refinement_keeps_reducing_the_fp64_residual(mg, tmp_path):
    """Property at a size the numpy restatement does not reach: every correction lowers the fp64
    residual of the fp64 iterate (the reference's cycle, undamped Jacobi + sampled restriction, is a
    slow iteration -- the contraction is modest, but it is monotone and it does not stall at the
    fp32 round-off level), and the analytic error of the refined iterate beats the single cycle's."""
    N = 2048
    path = str(tmp_path / "c.txt")
    mg.write_vcycle_file(path, N, 8, 3, 1e-7)
    single = mg.CyclePlan(path, fused=True, mixed=True).execute()
    plan = mg.CyclePlan(path, fused=True, mixed=True, refinement=12)
    got = plan.execute()
    assert got["status"] == 0 and single["status"] == 0
    e = got["refinement_errors"]
    assert len(e) == 11
    assert all(b < a for a, b in zip(e, e[1:])), e
    assert e[-1] < 0.2 * e[0], e
    assert got["mg_error"] < single["mg_error"]
    plan.close()


def test_command_line_program_mixed(mg, cycle_dir, tmp_path):
    """MG_MIXED=1 / MG_REFINE=k on the reference's command line: same report layout and CSV format,
    values within fp32 rounding of the fp64 run of the same file."""
    import os
    import shutil
    import subprocess
    shutil.copy(os.path.join(cycle_dir, "Vcycle.txt"), tmp_path)
    runs = {}
    for tag, env in (("f64", {}), ("mixed", {"MG_MIXED": "1"}), ("refine", {"MG_REFINE": "3"})):
        out = subprocess.run([mg.EXE_PATH, "4", "Vcycle.txt"], cwd=tmp_path, capture_output=True, text=True,
                             env=dict(os.environ, **env))
        assert out.returncode == 0, out.stdout + out.stderr
        assert "Output file name = Sol_HIP_Vcycle.txt" in out.stdout and "===== Final Result =====" in out.stdout
        runs[tag] = np.loadtxt(tmp_path / "Sol_HIP_Vcycle.txt", delimiter=",")
    scale = np.abs(runs["f64"]).max()
    assert runs["mixed"].shape == runs["f64"].shape
    assert 0 < np.abs(runs["mixed"] - runs["f64"]).max() <= 2e-5 * scale
    assert np.abs(runs["refine"] - runs["f64"]).max() <= 0.5 * scale   # another iterate, same problem


@pytest.mark.parametrize("R,collapse", [(2, 64), (3, 256), (8, 128)])
def test_mixed_slabs_virtual_ranks_vs_numpy(mg, oracle, tmp_path, R, collapse):
    """fp32 row slabs (virtual ranks on one GPU): the result and the errors are those of the
    single-GPU fp32 cycle, i.e. of the numpy restatement, bit for bit."""
    N = 1024
    path = str(tmp_path / "V.txt")
    mg.write_vcycle_file(path, N, 8, 3, 1e-7)
    toks = open(path).read().split()
    sizes, n = [], N
    while n >= 8:
        sizes.append(n)
        n //= 2
    U32, recs = o32.run_cycle_tokens(mg, oracle, oracle.getSource(N), 1.0, 3, sizes, toks[7:])
    plan = mg.SlabPlan(path, R, -1, collapse, mixed=True)
    for _ in range(2):
        got = plan.execute()
        assert got["status"] == 0
        assert np.array_equal(plan.gather_U(N), U32.astype(np.float64))
        assert [(g[0], g[1]) for g in got["records"]] == [(w[0], w[1]) for w in recs]
        for g, w in zip(got["records"], recs):
            assert g[3] == pytest.approx(w[2], rel=1e-10, abs=1e-300)
    single = mg.CyclePlan(path, fused=True, mixed=True).execute()
    assert got["mg_error"] == pytest.approx(single["mg_error"], rel=1e-10)
    plan.close()


@pytest.mark.parametrize("R,N,collapse,cycles", [(3, 1024, 128, 3), (8, 2048, 256, 2)])
def test_mixed_slab_refinement_matches_single_gpu(mg, tmp_path, R, N, collapse, cycles):
    """fp64 residual + correction between fp32 cycles on row slabs (one ghost exchange of the fp64
    iterate and one of the new fp32 source per extra cycle): the iterate and the residual norms of
    the single-GPU refinement (itself pinned to the numpy restatement), bit for bit / to 1e-12."""
    path = str(tmp_path / "V.txt")
    mg.write_vcycle_file(path, N, 8, 3, 1e-7)
    one = mg.CyclePlan(path, fused=True, mixed=True, refinement=cycles)
    want = one.execute(fetch_U=True)
    plan = mg.SlabPlan(path, R, -1, collapse, mixed=True, refinement=cycles)
    for _ in range(2):   # the second window has to restore the rounded source first
        got = plan.execute()
        assert got["status"] == 0 and want["status"] == 0
        assert np.array_equal(plan.gather_U(N), want["U"])
        assert got["refinement_errors"] == pytest.approx(want["refinement_errors"], rel=1e-12)
        assert got["mg_error"] == pytest.approx(want["mg_error"], rel=1e-10)
    plan.close(); one.close()
