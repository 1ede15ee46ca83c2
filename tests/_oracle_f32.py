"""numpy restatement of the MIXED-PRECISION mode's fp32 cycle (TEST INFRASTRUCTURE).

Parity unpinned: the reference's fp32 path is its CUDA file (src/MG_solver_GPU.cu), which cannot
be built or run here (no nvcc), and it uses different formulas (direct-form Jacobi, another error
metric, floorf/fmodf indices).  This oracle therefore restates the ENGINE's own definition of the
mode -- the fp64 operators of the CPU reference with every field and arithmetic operation in
fp32, transfer weights = the fp64 host tables rounded to fp32, norms accumulated in fp64 -- and
the tests hold the HIP kernels to it bit for bit; against the fp64 reference the mode is held to
a stated tolerance.  numpy float32 ufuncs round every operation to fp32 (no FMA contraction)."""
import numpy as np

f32 = np.float32


def spacings(N, L):
    dx = L / float(N - 1)
    return f32(dx * dx), f32(1.0 / (dx * dx))


def star(U):
    c = U[1:-1, 1:-1]
    return (((U[2:, 1:-1] + U[:-2, 1:-1]) + U[1:-1, 2:]) + U[1:-1, :-2]) - f32(4) * c


def sweep(U, F, dx2):
    out = U.copy()
    c = U[1:-1, 1:-1]
    out[1:-1, 1:-1] = c + f32(0.25) * (star(U) - dx2 * F[1:-1, 1:-1])
    return out


def residual(U, F, inv):
    D = np.zeros_like(U)
    D[1:-1, 1:-1] = inv * star(U) - F[1:-1, 1:-1]
    return D


def smoothing_error(U, F, inv):
    N = U.shape[0]
    r = np.abs(residual(U, F, inv)).astype(np.float64)
    rows, cols = np.indices(U.shape)
    mask = ((rows + cols) % 2 == 0)
    mask[0, :] = mask[-1, :] = mask[:, 0] = mask[:, -1] = False
    s = r[mask].sum()
    return (s + s) / N / N


def smooth(U, F, steps, L):
    dx2, inv = spacings(U.shape[0], L)
    for _ in range(steps):
        U = sweep(U, F, dx2)
    return U, smoothing_error(U, F, inv)


def restrict_neg_residual(mgmod, U, F, L, M):
    """doRestriction(N, -getResidual(U, F), M) in fp32 (weights rounded from the fp64 tables)."""
    N = U.shape[0]
    _, inv = spacings(N, L)
    D = np.negative(residual(U, F, inv))          # rim becomes -0.0 like the driver's sign flip
    lo, w = mgmod.restriction_table(N, M)
    a = w.astype(f32)
    b = f32(1) - a
    out = np.zeros((M, M), dtype=f32)
    rc, cc = np.arange(1, M - 1), np.arange(1, M - 1)
    fr, fc = lo[rc][:, None], lo[cc][None, :]
    A, B = a[cc][None, :], b[cc][None, :]
    C, Dw = a[rc][:, None], b[rc][:, None]         # c, d of the reference (rows)
    u0, u1, u2, u3 = D[fr, fc], D[fr, fc + 1], D[fr + 1, fc], D[fr + 1, fc + 1]
    out[1:-1, 1:-1] = B * Dw * u0 + A * Dw * u1 + C * B * u2 + A * C * u3
    return out


def prolong_add(mgmod, Uc, Uf):
    Nc, N = Uc.shape[0], Uf.shape[0]
    orow, rhi, rlo = mgmod.prolongation_table(Nc, N, 0)
    ocol, chi, clo = mgmod.prolongation_table(Nc, N, 1)
    I, J = np.meshgrid(orow, ocol, indexing="ij")
    c1, c2, c3, c4 = Uc[I, J], Uc[I, J + 1], Uc[I + 1, J], Uc[I + 1, J + 1]
    xh, xl = chi.astype(f32)[None, :], clo.astype(f32)[None, :]
    yh, yl = rhi.astype(f32)[:, None], rlo.astype(f32)[:, None]
    c_dx = f32(1.0 / float(Nc - 1))
    P = ((c1 * xh + c2 * xl) * yh + (c3 * xh + c4 * xl) * yl) / c_dx / c_dx
    return Uf + P


def gauss_seidel(oracle, F, L, tol):
    """The exact solver stays fp64 in the mixed mode: F widened exactly, the reference's own
    red-black Gauss-Seidel (oracle doExactSolver, src/MG_solver_CPU.cpp:952-1066), U rounded once."""
    N = F.shape[0]
    return oracle.doExactSolver(N, L, F.astype(np.float64), tol, 1).astype(f32)


def run_cycle_tokens(mgmod, oracle, F64, L, steps, sizes, tokens):
    """The fp32 cycle over a node stream (fixed steps, halving sizes).  Returns (U32 of the finest
    level, list of (node, N, error))."""
    levels = [dict(N=sizes[0], F=F64.astype(f32), U=None)]
    at, i, records = 0, 0, []
    while i < len(tokens):
        node = int(float(tokens[i])); i += 1
        if node == 2:
            break
        lv = levels[-1]
        if node == -1:
            at += 1
            lv["U"], e = smooth(np.zeros((lv["N"], lv["N"]), dtype=f32), lv["F"], steps, L)
            records.append((-1, lv["N"], e))
            M = sizes[at]
            levels.append(dict(N=M, F=restrict_neg_residual(mgmod, lv["U"], lv["F"], L, M), U=None))
        elif node == 0:
            tol = float(tokens[i]); i += 2
            lv["U"] = gauss_seidel(oracle, lv["F"], L, tol)
            records.append((0, lv["N"], 0.0))
        elif node == 1:
            at -= 1
            coarse = levels.pop()
            fine = levels[-1]
            fine["U"], e = smooth(prolong_add(mgmod, coarse["U"], fine["U"]), fine["F"], steps, L)
            records.append((1, fine["N"], e))
    return levels[0]["U"], records


def refine(mgmod, oracle, F64, L, steps, sizes, tokens, cycles):
    """`cycles` fp32 runs of the node stream joined by the fp64 residual of the fp64 iterate and an
    fp64 correction (mg_cycle_set_refinement).  Returns (U64, errors of the iterates 1..cycles-1)."""
    N = sizes[0]
    U = None
    errs = []
    for it in range(cycles):
        if it == 0:
            src = F64
        else:
            D = oracle.getResidual(N, L, U, F64)          # A U - F, rim 0
            rows, cols = np.indices(D.shape)
            even = ((rows + cols) % 2 == 0)
            s = np.abs(D[even]).sum()
            errs.append((s + s) / N / N)
            src = -D
        e32, _ = run_cycle_tokens(mgmod, oracle, src, L, steps, sizes, tokens)
        U = e32.astype(np.float64) if it == 0 else U + e32.astype(np.float64)
    return U, errs
