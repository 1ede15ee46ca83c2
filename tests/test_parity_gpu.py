"""Parity of the HIP operators with the CPU oracle, through the C ABI, on a real MI355X.
Bar (BASELINE.json north_star): restriction/prolongation indexing bit-exact; here every
output ARRAY is asserted bit-exact (the kernels keep the reference's evaluation order and
are built with -ffp-contract=off); scalar norms, whose summation order is unspecified even
in the reference (OpenMP reduction), are held to 1e-12 relative."""
import os

import numpy as np
import pytest

import _synth
from conftest import assert_bits

pytestmark = pytest.mark.gpu

REL = 1e-12  # tolerance for scalar norms (reduction order)
SIZES = [3, 4, 5, 8, 15, 16, 17, 33, 64, 100, 129, 257, 512, 1000]
SMOOTHERS = ["stream", "simple"]


def rand_pair(N, seed):
    rng = np.random.default_rng(seed)
    return rng.random((N, N)), rng.random((N, N)) - 0.5


@pytest.fixture(params=SMOOTHERS)
def smoother(request, mg):
    mg.set_smoother(request.param)
    yield request.param
    mg.set_smoother("stream")


@pytest.mark.parametrize("N", SIZES)
def test_smoothing_vs_oracle(mg, oracle, smoother, N):
    U0, F = rand_pair(N, N)
    Fd = mg.DeviceGrid.from_host(F)
    for step in (1, 2, 3, 4, 5, 10):
        U = mg.DeviceGrid.from_host(U0)
        err = mg.doSmoothing(N, 1.0, U, Fd, step)
        want, werr = oracle.doSmoothing(N, 1.0, U0, F, step)
        assert_bits(U.to_host(), want, f"doSmoothing N={N} step={step} ({smoother})")
        assert err == pytest.approx(werr, rel=REL), (N, step)


@pytest.mark.parametrize("N", [16, 64, 257, 1024])
@pytest.mark.parametrize("L", [1.0, 2.5])
def test_smoothing_fused_forms(mg, oracle, smoother, N, L):
    """mg_smooth_pp: zero start (memset folded in), fused error, fused +-residual."""
    U0, F = rand_pair(N, 7 * N)
    Fd = mg.DeviceGrid.from_host(F)
    for step in (1, 3, 6):
        out, D = mg.DeviceGrid(N), mg.DeviceGrid(N)
        err = mg.smooth_pp(N, L, None, out, Fd, step, want_error=True, D_out=D, d_sign=-1)
        want, werr = oracle.doSmoothing(N, L, np.zeros((N, N)), F, step)
        assert_bits(out.to_host(), want, f"zero-start smoothing N={N} step={step}")
        assert err == pytest.approx(werr, rel=REL)
        assert_bits(D.to_host(), -oracle.getResidual(N, L, want, F), "fused -residual")
        Uin = mg.DeviceGrid.from_host(U0)
        mg.smooth_pp(N, L, Uin, out, Fd, step, D_out=D, d_sign=+1)
        want, _ = oracle.doSmoothing(N, L, U0, F, step)
        assert_bits(out.to_host(), want, "out-of-place smoothing")
        assert_bits(D.to_host(), oracle.getResidual(N, L, want, F), "fused +residual")


# (66 ... 1022: sizes of the register-tile kernel -- levels 65...1024 -- that are no multiple of its tile, incl. coarse sizes that are odd)
@pytest.mark.parametrize("N,M", [(8, 4), (16, 8), (64, 32), (100, 50), (256, 128), (1024, 512), (2048, 1024),
                                  (33, 16), (16, 15), (100, 37), (250, 124), (66, 33), (72, 36), (130, 65), (514, 257), (1000, 500), (1022, 511)])
@pytest.mark.parametrize("step", [1, 2, 3, 4, 6])
def test_fused_smooth_restrict_vs_oracle(mg, oracle, smoother, N, M, step):
    """One "-1" node (src/MG_solver_CPU.cpp:252-287): [zero U,] smooth, residual, sign flip,
    restrict -- fused into one pass where the streaming kernel can, operator by operator
    otherwise (odd N, non-nested sizes, simple smoother); always the oracle's bits."""
    U0, F = rand_pair(N, 11 * N + M)
    Fd = mg.DeviceGrid.from_host(F)
    for zero in (True, False):
        start = np.zeros((N, N)) if zero else U0
        want_U, want_err = oracle.doSmoothing(N, 1.0, start, F, step)
        want_Fc = oracle.doRestriction(N, -oracle.getResidual(N, 1.0, want_U, F), M)
        out = mg.DeviceGrid.from_host(np.full((N, N), np.nan))
        Fc = mg.DeviceGrid.from_host(np.full((M, M), np.nan))
        Uin = None if zero else mg.DeviceGrid.from_host(U0)
        err = mg.smooth_restrict(N, 1.0, Uin, out, Fd, step, M, Fc, want_error=True)
        assert_bits(out.to_host(), want_U, f"smooth_restrict U N={N} step={step} zero={zero}")
        assert_bits(Fc.to_host(), want_Fc, f"smooth_restrict F_coarse {N}->{M} step={step} zero={zero}")
        assert err == pytest.approx(want_err, rel=REL)


@pytest.mark.parametrize("Nc,N", [(4, 8), (8, 16), (32, 64), (50, 100), (128, 256), (512, 1024), (1024, 2048),
                                   (16, 33), (15, 16), (37, 100), (124, 250), (33, 66), (36, 72), (65, 130), (257, 514), (500, 1000), (511, 1022)])
@pytest.mark.parametrize("step", [1, 2, 3, 4, 6])
def test_fused_prolong_smooth_vs_oracle(mg, oracle, smoother, Nc, N, step):
    """One "1" node (src/MG_solver_CPU.cpp:353-416): prolong, add, smooth."""
    rng = np.random.default_rng(13 * N + Nc)
    Uc, Uf, F = rng.random((Nc, Nc)) - 0.5, rng.random((N, N)), rng.random((N, N)) - 0.5
    want0 = oracle.doGridAddition(N, Uf, oracle.doProlongation(Nc, Uc, N, fill=0.0))
    want, want_err = oracle.doSmoothing(N, 1.0, want0, F, step)
    out = mg.DeviceGrid.from_host(np.full((N, N), np.nan))
    err = mg.prolong_smooth(Nc, mg.DeviceGrid.from_host(Uc), N, 1.0, mg.DeviceGrid.from_host(Uf), out,
                            mg.DeviceGrid.from_host(F), step, want_error=True)
    assert_bits(out.to_host(), want, f"prolong_smooth {Nc}->{N} step={step}")
    assert err == pytest.approx(want_err, rel=REL)


@pytest.mark.parametrize("N", [4096, 8192])
def test_fullsize_fused_nodes_match_unfused(mg, N):
    """At benchmark size the fused node kernels must reproduce the operator-by-operator
    sequence (itself pinned to the reference by the checksum tests) bit for bit."""
    M = N // 2
    F = mg.DeviceGrid.uniform(N, 7)
    # down: zero start
    U1, D, Fc1 = mg.DeviceGrid(N), mg.DeviceGrid(N), mg.DeviceGrid(M)
    mg.smooth_pp(N, 1.0, None, U1, F, 3, D_out=D, d_sign=-1)
    mg.doRestriction(N, D, M, Fc1)
    U2, Fc2 = mg.DeviceGrid(N), mg.DeviceGrid(M)
    mg.smooth_restrict(N, 1.0, None, U2, F, 3, M, Fc2)
    assert U1.checksum() == U2.checksum()
    assert Fc1.checksum() == Fc2.checksum()
    # up
    Uc = mg.DeviceGrid.uniform(M, 8)
    mg.prolongAdd(M, Uc, N, U1, D)          # D = U1 + P(Uc)
    mg.smooth_pp(N, 1.0, D, U2, F, 3)       # U2 = smooth^3(D)
    O = mg.DeviceGrid(N)
    mg.prolong_smooth(M, Uc, N, 1.0, U1, O, F, 3)
    assert O.checksum() == U2.checksum()
    for g in (F, U1, D, Fc1, U2, Fc2, Uc, O):
        g.free()
    mg.lib().mg_pool_trim()


@pytest.mark.parametrize("N", SIZES)
def test_residual_add_negate_vs_oracle(mg, oracle, N):
    U0, F = rand_pair(N, 3 * N)
    U, Fd, D = mg.DeviceGrid.from_host(U0), mg.DeviceGrid.from_host(F), mg.DeviceGrid(N)
    mg.getResidual(N, 1.7, U, Fd, D)
    want = oracle.getResidual(N, 1.7, U0, F)
    assert_bits(D.to_host(), want, "getResidual")
    mg.negate(N, D)
    assert_bits(D.to_host(), -want, "negate")   # rim becomes -0.0 exactly as in the reference
    mg.doGridAddition(N, U, Fd)
    assert_bits(U.to_host(), oracle.doGridAddition(N, U0, F), "doGridAddition")


PAIRS_R = [(16, 8), (17, 9), (16, 15), (33, 16), (64, 32), (100, 37), (257, 128), (512, 256), (1000, 500), (1024, 512), (2048, 1024)]
PAIRS_P = [(4, 8), (8, 16), (15, 16), (9, 17), (16, 33), (32, 64), (37, 100), (128, 257), (256, 512), (500, 1000), (1024, 2048)]


@pytest.mark.parametrize("N,M", PAIRS_R)
def test_restriction_vs_oracle(mg, oracle, N, M):
    Uf = np.random.default_rng(N + M).random((N, N)) - 0.3
    d_f, d_c = mg.DeviceGrid.from_host(Uf), mg.DeviceGrid.from_host(np.full((M, M), np.nan))
    mg.doRestriction(N, d_f, M, d_c)
    want = oracle.doRestriction(N, Uf, M)
    assert_bits(d_c.to_host(), want, f"doRestriction {N}->{M}")
    # the driver's sequence negate-then-restrict, and the folded form
    mg.restrict_signed(N, d_f, M, d_c, -1)
    assert_bits(d_c.to_host(), oracle.doRestriction(N, -Uf, M), "restrict(-x) folded", zero_sign=True)


@pytest.mark.parametrize("N,M", PAIRS_P)
def test_prolongation_vs_oracle(mg, oracle, N, M):
    Uc = np.random.default_rng(N * M).random((N, N)) - 0.3
    d_c, d_f = mg.DeviceGrid.from_host(Uc), mg.DeviceGrid.from_host(np.zeros((M, M)))
    mg.doProlongation(N, d_c, M, d_f)
    want = oracle.doProlongation(N, Uc, M, fill=0.0)
    assert_bits(d_f.to_host(), want, f"doProlongation {N}->{M}")
    base = np.random.default_rng(1).random((M, M))
    d_b, d_o = mg.DeviceGrid.from_host(base), mg.DeviceGrid(M)
    mg.prolongAdd(N, d_c, M, d_b, d_o)
    assert_bits(d_o.to_host(), oracle.doGridAddition(M, base, want), "prolong+add fused")


@pytest.mark.parametrize("N", [4, 5, 8, 15, 16, 17, 32, 33, 64, 96, 100, 128])
def test_exact_solver_vs_oracle(mg, oracle, N):
    F = np.random.default_rng(N).random((N, N)) - 0.5
    tol = 1e-7 if N <= 33 else 1e-3
    Fd, U = mg.DeviceGrid.from_host(F), mg.DeviceGrid.from_host(np.full((N, N), 3.0))
    mg.doExactSolver(N, 1.0, U, Fd, tol, 1)
    want = oracle.doExactSolver(N, 1.0, F, tol)
    assert mg.lastExactSolverIterations() == oracle.gs_iterations()
    assert_bits(U.to_host(), want, f"GaussSeidel N={N}")


@pytest.mark.parametrize("N", [4, 6, 8])
@pytest.mark.parametrize("scale,tol", [(0.0, 1e-7), (1e-6, 1e-7), (1.0, 1e-7), (1.0, 1e-3), (1.0, 1e-11), (1e2, 1e-7), (1e3, 1e-7),
                                       (1e4, 1e-7), (1e9, 1e-2), (1e-3, 1e-12)])
def test_block_exact_solver_scales_and_targets(mg, oracle, N, scale, tol):
    """The block solver of the coarse tail (even grids up to 8 x 8) judges most iterates without the norm, in batches,
    under a bound that depends on max|F| / tolerance: right-hand sides from zero to 1e9 walk through every path of it
    (all batches certain; the stop inside the first batch; the bound exhausted after a few sweeps, so that the norm is
    computed on every iterate) -- U bit for bit and the iteration count against the oracle."""
    F = (np.random.default_rng(100 * N + int(np.log10(scale + 1e-30)) % 97).random((N, N)) - 0.5) * scale
    Fd, U = mg.DeviceGrid.from_host(F), mg.DeviceGrid.from_host(np.full((N, N), 3.0))
    mg.doExactSolver(N, 1.0, U, Fd, tol, 1)
    want = oracle.doExactSolver(N, 1.0, F, tol)
    assert mg.lastExactSolverIterations() == oracle.gs_iterations()
    assert_bits(U.to_host(), want, f"block GaussSeidel N={N} scale={scale} tol={tol}")


def _gs_norms(N, F, iters):
    """red-black Gauss-Seidel in numpy (the oracle's formulas, src/MG_solver_CPU.cpp:993-1059): the norm of every iterate,
    good to a few ulp -- only used to PLACE tolerances next to the norm of a chosen iterate."""
    h2 = (1.0 / (N - 1)) ** 2
    inv = 1.0 / h2
    U = np.zeros((N, N))
    r, c = np.meshgrid(np.arange(N), np.arange(N), indexing="ij")
    inner = (r > 0) & (r < N - 1) & (c > 0) & (c < N - 1)
    out = []
    for _ in range(iters):
        for colour in (0, 1):
            m = inner & (((r + c) & 1) == colour)
            nb = np.zeros((N, N))
            nb[1:-1, 1:-1] = U[1:-1, :-2] + U[1:-1, 2:] + U[2:, 1:-1] + U[:-2, 1:-1]
            U[m] = 0.25 * (nb[m] - h2 * F[m])
        res = np.zeros((N, N))
        res[1:-1, 1:-1] = inv * (U[2:, 1:-1] + U[:-2, 1:-1] + U[1:-1, 2:] + U[1:-1, :-2] - 4 * U[1:-1, 1:-1]) - F[1:-1, 1:-1]
        out.append(np.abs(res).sum() / ((N - 2) ** 2))
    return out


@pytest.mark.parametrize("kind", ["point", "checkerboard", "constant", "coarse_residual", "alternating_rows"])
def test_block_exact_solver_stops_on_every_sweep_of_a_batch(mg, oracle, kind):
    """Round-2 advisor finding: the block solver skips the reference's per-sweep `err > target` test for 8 sweeps at a
    time and reconstructs the stopping iterate afterwards; the only pin were random right-hand sides.  Here: structured
    right-hand sides (a single point, a checkerboard, a constant, alternating rows, and the 8 x 8 right-hand side a
    V-cycle actually hands the coarse solve) with the tolerance placed a hair above and below the norm of a chosen
    iterate (2^-30 and 2^-12: inside the 2^-10 band in which the solver computes the norm itself; 2^-9: outside it) --
    iterates 1, 7, 8, 9, 12, 15, 16, 17, 24, 25: the first, last and middle sweep of a batch, so the stop falls on every
    position relative to the batch boundaries.  Iteration count and U, bit for bit, against the oracle."""
    N = 8
    r, c = np.meshgrid(np.arange(N), np.arange(N), indexing="ij")
    if kind == "point":
        F = np.zeros((N, N)); F[3, 4] = 1.0
    elif kind == "checkerboard":
        F = np.where((r + c) % 2 == 0, 1.0, -1.0)
    elif kind == "constant":
        F = np.full((N, N), -2.5)
    elif kind == "alternating_rows":
        F = np.where(r % 2 == 0, 3.0, -1.0) * 1e-3
    else:
        # what the coarse solve of a V(3,3)-cycle sees: the restricted residuals of getSource down 64 -> 8
        Fl, n = oracle.getSource(64, 1.0, 0.0, 0.0), 64
        while n > N:
            Ul, _ = oracle.doSmoothing(n, 1.0, np.zeros((n, n)), Fl, 3)
            Fl = oracle.doRestriction(n, -oracle.getResidual(n, 1.0, Ul, Fl), n // 2)
            n //= 2
        F = Fl
    F = np.ascontiguousarray(F, dtype=np.float64)
    F[0, :] = F[-1, :] = 0.0; F[:, 0] = F[:, -1] = 0.0  # (the rim of a right-hand side is never read)
    norms = _gs_norms(N, F, 26)
    Fd = mg.DeviceGrid.from_host(F)
    checked = 0
    for k in (1, 7, 8, 9, 12, 15, 16, 17, 24, 25):
        base = norms[k - 1]
        if not (base > 1e-300):
            continue
        # (not AT the norm: within a few ulp of it the verdict depends on the order in which the 36 residuals are summed,
        # which the reference leaves to its OpenMP reduction -- 2^-30 is a million times that and a million times less
        # than the solver's 2^-10 band)
        for f in (1 - 2.0 ** -30, 1 + 2.0 ** -30, 1 - 2.0 ** -12, 1 + 2.0 ** -12, 1 - 2.0 ** -9, 1 + 2.0 ** -9):
            tol = base * f
            U = mg.DeviceGrid.from_host(np.full((N, N), 7.0))
            mg.doExactSolver(N, 1.0, U, Fd, tol, 1)
            want = oracle.doExactSolver(N, 1.0, F, tol)
            assert mg.lastExactSolverIterations() == oracle.gs_iterations(), (kind, k, f)
            assert_bits(U.to_host(), want, f"block GaussSeidel {kind} tol next to iterate {k} (x{f})")
            checked += 1
    assert checked >= 30


def test_exact_solver_multi_workgroup_path(mg, oracle):
    N = 160  # above the single-workgroup LDS limit
    F = np.random.default_rng(5).random((N, N)) - 0.5
    Fd, U = mg.DeviceGrid.from_host(F), mg.DeviceGrid(N)
    mg.doExactSolver(N, 1.0, U, Fd, 5e-3, 1)
    want = oracle.doExactSolver(N, 1.0, F, 5e-3)
    assert mg.lastExactSolverIterations() == oracle.gs_iterations()
    assert_bits(U.to_host(), want, "GaussSeidel multi-workgroup")


def test_exact_solver_option0_refused(mg):
    """src/MG_solver_GPU.cu:1286-1289: the GPU build refuses the inverse-matrix solver."""
    U, F = mg.DeviceGrid.zeros(8), mg.DeviceGrid.zeros(8)
    with pytest.raises(mg.MGError, match="Inverse Matrix"):
        mg.doExactSolver(8, 1.0, U, F, 1e-7, 0)


def test_problem_definition(mg, oracle):
    for N in (16, 129):
        F = mg.getSource(N, 1.5, 0.25, -0.5)
        assert_bits(F.to_host(), oracle.getSource(N, 1.5, 0.25, -0.5), "getSource (host libm form)")
        A = mg.getAnalytic(N).to_host()
        np.testing.assert_allclose(A, oracle.getAnalytic(N), rtol=2e-15, atol=0)  # device exp: a few ulp after the products
        U = np.random.default_rng(N).random((N, N))
        got = mg.analyticError(N, 1.0, mg.DeviceGrid.from_host(U))
        want = np.abs(oracle.getAnalytic(N) - U).sum() / (N * N)
        assert got == pytest.approx(want, rel=1e-12)


# ------------------------------------------------------------------ golden vectors
@pytest.mark.parametrize("N", [16, 64])
def test_golden_ops(mg, golden_ops, N):
    g = golden_ops
    for tag, U0, F in (("src", np.zeros((N, N)), g[f"N{N}_F_source"]), ("rnd", g[f"N{N}_U_rand"], g[f"N{N}_F_rand"])):
        Fd = mg.DeviceGrid.from_host(F)
        for s in (1, 3, 10):
            U = mg.DeviceGrid.from_host(U0)
            e = mg.doSmoothing(N, 1.0, U, Fd, s)
            assert_bits(U.to_host(), g[f"N{N}_{tag}_smooth{s}_U"], f"golden smoothing {tag} {s}")
            assert e == pytest.approx(g[f"N{N}_{tag}_smooth{s}_err"][0], rel=REL)
        U3 = mg.DeviceGrid.from_host(g[f"N{N}_{tag}_smooth3_U"])
        D = mg.DeviceGrid(N)
        mg.getResidual(N, 1.0, U3, Fd, D)
        assert_bits(D.to_host(), g[f"N{N}_{tag}_residual"], "golden residual")
        mg.negate(N, D)
        C = mg.DeviceGrid(N // 2)
        mg.doRestriction(N, D, N // 2, C)
        assert_bits(C.to_host(), g[f"N{N}_{tag}_restrict_negD"], "golden restrict(-D)")
        half = mg.DeviceGrid.from_host(g[f"N{N}_{tag}_smooth3_U"][: N // 2, : N // 2].copy())
        P = mg.DeviceGrid(N)
        mg.doProlongation(N // 2, half, N, P)
        assert_bits(P.to_host(), g[f"N{N}_{tag}_prolong_from_half"], "golden prolong")
        mg.doGridAddition(N, U3, Fd)
        assert_bits(U3.to_host(), g[f"N{N}_{tag}_add"], "golden add")


def test_golden_odd_pairs_kats_and_gs(mg, golden_ops):
    g = golden_ops
    for (Nf, Mc) in ((17, 9), (16, 15), (33, 16)):
        out = mg.DeviceGrid(Mc)
        mg.doRestriction(Nf, mg.DeviceGrid.from_host(g[f"restrict_{Nf}to{Mc}_in"]), Mc, out)
        assert_bits(out.to_host(), g[f"restrict_{Nf}to{Mc}_out"], f"golden restrict {Nf}->{Mc}")
    for (Nc, Mf) in ((15, 16), (9, 17), (16, 33)):
        out = mg.DeviceGrid.zeros(Mf)
        mg.doProlongation(Nc, mg.DeviceGrid.from_host(g[f"prolong_{Nc}to{Mf}_in"]), Mf, out)
        assert_bits(out.to_host(), g[f"prolong_{Nc}to{Mf}_out"], f"golden prolong {Nc}->{Mf}")
    # the reference's own analytic KATs (testFunction/Test_doRestriction_GPU.cu:189-193,
    # testFunction/Test_doProlongation_GPU.cu:190-194)
    out = mg.DeviceGrid(8)
    mg.doRestriction(16, mg.DeviceGrid.from_host(np.add.outer(np.arange(16.0), np.arange(16.0))), 8, out)
    assert_bits(out.to_host(), g["kat_restrict_16to8"], "KAT restrict")
    out = mg.DeviceGrid.zeros(8)
    mg.doProlongation(4, mg.DeviceGrid.from_host(np.add.outer(np.arange(4.0), np.arange(4.0))), 8, out)
    assert_bits(out.to_host(), g["kat_prolong_4to8"], "KAT prolong")
    for N in (8, 16, 17):
        U = mg.DeviceGrid(N)
        mg.doExactSolver(N, 1.0, U, mg.DeviceGrid.from_host(g[f"gs_N{N}_F"]), 1e-7, 1)
        assert_bits(U.to_host(), g[f"gs_N{N}_U"], f"golden GaussSeidel {N}")


def test_reference_test_program_configurations(mg, oracle):
    """The shapes the reference's own hand-run comparison programs use:
    Test_doSmoothing_GPU.cu (N=16, rand U/F, 10 steps), Test_getResidual_GPU.cu (N=16),
    Test_doExactSolver_GPU_Double.cu (N=16, tol 1e-3, option 1) and Test_doProlongation_GPU.cu,
    which UPSAMPLES with doRestriction used as a generic zoom (N=4 -> M=8, :233)."""
    N = 16
    rng = np.random.default_rng(2020)
    U0, F = rng.random((N, N)), rng.random((N, N))  # rand()/RAND_MAX: uniform [0,1]
    U, Fd = mg.DeviceGrid.from_host(U0), mg.DeviceGrid.from_host(F)
    err = mg.doSmoothing(N, 1.0, U, Fd, 10)
    want, werr = oracle.doSmoothing(N, 1.0, U0, F, 10)
    assert_bits(U.to_host(), want, "Test_doSmoothing shape")
    assert err == pytest.approx(werr, rel=REL)
    D = mg.DeviceGrid(N)
    mg.getResidual(N, 1.0, mg.DeviceGrid.from_host(U0), Fd, D)
    assert_bits(D.to_host(), oracle.getResidual(N, 1.0, U0, F), "Test_getResidual shape")
    G = mg.DeviceGrid.from_host(U0)
    mg.doExactSolver(N, 1.0, G, Fd, 1e-3, 1)
    assert_bits(G.to_host(), oracle.doExactSolver(N, 1.0, F, 1e-3), "Test_doExactSolver shape")
    ramp4 = np.add.outer(np.arange(4.0), np.arange(4.0))
    out = mg.DeviceGrid.from_host(np.full((8, 8), np.nan))
    mg.doRestriction(4, mg.DeviceGrid.from_host(ramp4), 8, out)   # "restriction" 4 -> 8: a zoom
    zoom = oracle.doRestriction(4, ramp4, 8)
    assert_bits(out.to_host(), zoom, "doRestriction as generic zoom 4->8")
    ix = np.arange(8.0)
    np.testing.assert_allclose(zoom[1:-1, 1:-1], (np.add.outer(ix, ix) * 3.0 / 7.0)[1:-1, 1:-1], rtol=1e-13)


def test_engine_on_a_torch_stream_and_tensors():
    """PyTorch is plumbing here (device memory, streams): the engine enqueues on a caller's
    hipStream and works on memory it did not allocate.  Runs in a subprocess because torch has to
    be imported before the engine library (one HIP runtime per process)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, "_torch_interop_worker.py")], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0 and "TORCH_INTEROP OK" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]


def test_engine_loaded_before_torch_exits_cleanly():
    """The other import order: the engine is loaded and used FIRST, torch is imported afterwards, the engine
    then works on torch tensors, and the process must end with exit code 0 with ONE HIP runtime mapped.
    (libmgpoisson.so no longer links RCCL -- it is resolved lazily inside mg_comm_init -- and the Python binding
    maps the installed torch wheel's own libamdhip64 before the engine, see _bind_hip_runtime; both used to be
    second copies of runtimes torch brings along, and two runtimes on one device abort at exit.)"""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = {k: v for k, v in os.environ.items() if k != "MG_HIP_RUNTIME"}
    out = subprocess.run([sys.executable, os.path.join(here, "_engine_first_worker.py")], capture_output=True, text=True,
                         timeout=600, env=env)
    assert out.returncode == 0 and "ENGINE_FIRST OK" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]


# ------------------------------------------------------------------ synthetic data
def test_synthetic_fill_and_checksum(mg):
    for N, seed in ((64, 11), (257, 22)):
        g = mg.DeviceGrid.uniform(N, seed)
        host = _synth.hash_field(N, seed)
        assert_bits(g.to_host(), host, "mg_fill_uniform")
        assert g.checksum() == _synth.checksum(host)
    z = mg.DeviceGrid.from_host(np.array([[0.0, -0.0], [1.5, -2.25]]))
    assert z.checksum() == _synth.checksum(np.array([[0.0, 0.0], [1.5, -2.25]]))


# ------------------------------------------------------------------ full benchmark sizes
@pytest.mark.parametrize("N", [4096, 8192, 16384])
def test_fullsize_smoothing_and_residual_checksums(mg, golden_fullsize, smoother, N):
    """BASELINE.json sizes (16384: the finest grid of configs 4 and 5): inputs are hash-generated on the
    device, the outputs' 128-bit checksums must equal those of the REFERENCE's doSmoothing/getResidual
    on the same inputs (tests/golden/make_golden.py --big / --add-16384)."""
    U, F, D = mg.DeviceGrid.uniform(N, 11), mg.DeviceGrid.uniform(N, 22), mg.DeviceGrid(N)
    mg.getResidual(N, 1.0, U, F, D)
    assert list(D.checksum()) == golden_fullsize[f"residual_N{N}"]["checksum"]
    err = mg.doSmoothing(N, 1.0, U, F, 3)
    assert list(U.checksum()) == golden_fullsize[f"smooth3_N{N}"]["checksum"]
    assert err == pytest.approx(golden_fullsize[f"smooth3_N{N}"]["error"], rel=REL)
    for g in (U, F, D):
        g.free()
    mg.lib().mg_pool_trim()


@pytest.mark.parametrize("N", [8192, 16384])
def test_fullsize_single_sweeps_reach_the_reference_checksum(mg, golden_fullsize, N):
    """north_star's yardstick kernel -- ONE Jacobi sweep per launch on a large grid (from N = 8192 on the multi-row form
    of the pair kernel: four rows per thread, rolling row window in registers) -- pinned to the REFERENCE: three single
    sweeps must give the checksum of the reference's doSmoothing(3) on the same hash-generated inputs, and one single
    sweep the bits of the streaming kernel at S = 1."""
    U, F, A, B = mg.DeviceGrid.uniform(N, 11), mg.DeviceGrid.uniform(N, 22), mg.DeviceGrid(N), mg.DeviceGrid(N)
    mg.smooth_pp(N, 1.0, U, A, F, 1)
    first = A.checksum()
    mg.smooth_pp(N, 1.0, A, B, F, 1)
    mg.smooth_pp(N, 1.0, B, A, F, 1)
    assert list(A.checksum()) == golden_fullsize[f"smooth3_N{N}"]["checksum"]
    mg.set_smoother("stream_only")
    try:
        mg.smooth_pp(N, 1.0, U, B, F, 1)
    finally:
        mg.set_smoother("stream")
    assert B.checksum() == first
    for g in (U, F, A, B):
        g.free()
    mg.lib().mg_pool_trim()


def ulp_distance(a, b):
    """distance in units in the last place between two fp64 arrays of the same sign pattern"""
    ia, ib = a.view(np.int64), b.view(np.int64)
    ia = np.where(ia < 0, np.int64(-2 ** 63) - ia, ia)  # map the sign-magnitude bit patterns onto a number line
    ib = np.where(ib < 0, np.int64(-2 ** 63) - ib, ib)
    return np.abs(ia - ib)


@pytest.mark.parametrize("N,L,mx,my", [(64, 1.5, 0.25, -0.5), (257, 1.0, 0.0, 0.0), (2048, 1.0, 0.0, 0.0), (4096, 3.0, -1.0, 2.0)])
def test_device_source_is_the_reference_source(mg, oracle, N, L, mx, my):
    """SURVEY.md section 8 row (f-1), src/MG_solver_CPU.cpp:468-493 (GPU twin MG_solver_GPU.cu:502-528): getSource on
    the device -- no host pass, no PCIe.  The reference calls libm's exp(); k_source evaluates glibc's algorithm for it
    (table from scripts/gen_exp_table.py, FMA form), so on a host with that libm F is the reference's bit for bit, and
    that is what makes the device form the default ("auto" mode runs this comparison itself on ~130k points).  On any
    other host the engine falls back to the host form; the device form is then still within 2 ulp."""
    want = oracle.getSource(N, L, mx, my)
    mg.set_source("host")
    try:
        host = mg.getSource(N, L, mx, my).to_host()
        mg.set_source("device")
        dev = mg.getSource(N, L, mx, my).to_host()
    finally:
        mg.set_source("auto")
    assert_bits(host, want, "host getSource == reference getSource")
    assert np.array_equal(dev[0], want[0]) and np.array_equal(dev[:, -1], want[:, -1])
    if mg.lib().mg_source_is_bit_identical():
        assert mg.source_mode() == "device"
        assert_bits(dev, want, "device getSource == reference getSource")
    else:
        assert mg.source_mode() == "host"
        assert ulp_distance(dev, want).max() <= 2


def test_device_exp_is_this_hosts_libm(mg):
    """The image's glibc (2.35, FMA variant selected on this CPU) is what exp_libm was written against: the check
    must pass here, otherwise the default silently went back to the host form."""
    assert mg.lib().mg_source_is_bit_identical() == 1 and mg.source_mode() == "device"


def test_cycle_with_device_source(mg, oracle, cycle_dir):
    """The same row end to end: a V-cycle whose F comes from k_source, against one whose F comes from the host's libm
    (bit-identical when the device form reproduces that libm, see above; within 1e-9 / 1e-10 otherwise)."""
    path = os.path.join(cycle_dir, "Vcycle.txt")
    want = oracle.run_cycle_file(path)
    mg.set_source("device")
    try:
        plan = mg.CyclePlan(path, fused=True)
        got = plan.execute(fetch_U=True)
        plan.close()
    finally:
        mg.set_source("auto")
    assert got["status"] == 0
    assert got["mg_error"] == pytest.approx(want["mg_error"], rel=1e-9)
    for g, w in zip(got["records"], want["records"]):
        assert tuple(g[:3]) == tuple(w[:3]) and g[3] == pytest.approx(w[3], rel=1e-10, abs=1e-300)
    np.testing.assert_allclose(got["U"], want["U"], rtol=1e-10, atol=1e-18)
    if mg.lib().mg_source_is_bit_identical():
        assert_bits(got["U"], want["U"], "V-cycle on the device source", zero_sign=True)


@pytest.mark.parametrize("N", [8192, 16384, 32768])
def test_fullsize_transfer_checksums(mg, golden_fullsize, N):
    """Bit-exact R/P indexing at 8192 .. 32768 against the reference itself."""
    M = N // 2
    Uf, Uc = mg.DeviceGrid.uniform(N, 33), mg.DeviceGrid(M)
    mg.doRestriction(N, Uf, M, Uc)
    assert list(Uc.checksum()) == golden_fullsize[f"restrict_{N}to{M}"]["checksum"]
    Uc.free()
    src = mg.DeviceGrid.uniform(M, 44)
    mg.lib().mg_fill_zero(Uf.ptr, Uf.size)
    mg.doProlongation(M, src, N, Uf)
    assert list(Uf.checksum()) == golden_fullsize[f"prolong_{M}to{N}"]["checksum"]
    Uf.free(); src.free()
    mg.lib().mg_pool_trim()


def test_large_grid_forms_of_the_elementwise_operators(mg):
    """From N = 4096 on the operator-by-operator kernels run their 16-byte non-temporal forms (k_residual_pairs, k_add_pairs,
    k_negate_pairs, k_prolong_pairs; the residual and the prolongation are pinned by the reference's checksums in the tests
    above): negation and addition against numpy on the same arrays, bit for bit."""
    N = 4096
    U, F, D, E = mg.DeviceGrid.uniform(N, 11), mg.DeviceGrid.uniform(N, 22), mg.DeviceGrid(N), mg.DeviceGrid(N)
    mg.getResidual(N, 1.0, U, F, D)
    d = D.to_host()
    mg.negate(N, D)
    assert_bits(D.to_host(), -d, "negate at 4096")
    u, f = U.to_host(), F.to_host()
    mg.doGridAddition(N, U, F)
    assert_bits(U.to_host(), u + f, "doGridAddition at 4096")
    for g in (U, F, D, E):
        g.free()
    mg.lib().mg_pool_trim()


@pytest.mark.parametrize("N", [8192])
def test_fullsize_properties(mg, smoother, N):
    """Size-independent properties at the headline size."""
    F0 = mg.DeviceGrid.zeros(N)
    # a constant field is a fixed point of the sweep when F = 0 (U + 0.25*(4U - 4U - 0))
    U = mg.DeviceGrid.from_host(np.full((N, N), 0.625))
    c0 = U.checksum()
    mg.doSmoothing(N, 1.0, U, F0, 3, want_error=False)
    assert U.checksum() == c0
    # 6 sweeps == 3 + 3 sweeps; zero start == explicit zeros
    F = mg.DeviceGrid.uniform(N, 5)
    A, B = mg.DeviceGrid.uniform(N, 6), mg.DeviceGrid.uniform(N, 6)
    mg.doSmoothing(N, 1.0, A, F, 6, want_error=False)
    mg.doSmoothing(N, 1.0, B, F, 3, want_error=False)
    mg.doSmoothing(N, 1.0, B, F, 3, want_error=False)
    assert A.checksum() == B.checksum()
    Z, O = mg.DeviceGrid.zeros(N), mg.DeviceGrid(N)
    mg.doSmoothing(N, 1.0, Z, F, 3, want_error=False)
    mg.smooth_pp(N, 1.0, None, O, F, 3)
    assert Z.checksum() == O.checksum()
    for g in (F0, U, F, A, B, Z, O):
        g.free()
    mg.lib().mg_pool_trim()
