"""World-size-N gloo worker (CPU): emulates the row-slab V-cycle of mg_slab.cpp with the CPU
oracle as the local operator.  Every rank holds full-size arrays that are NaN outside its
window (owned rows + GHOST halo rows), applies the oracle's whole-grid operators, exchanges
exactly the ghost rows the engine exchanges (on the way down only; nothing on the way up) (over torch.distributed/gloo) and finally the
assembled finest U must equal the oracle's single-domain result bit for bit with no NaN in
any owned row -- which proves the partition (mg_slab_partition, the product's host code), the
halo depth and the exchange schedule.  TEST INFRASTRUCTURE (uses oracle/).

torch is imported before the engine library on purpose (one HIP runtime per process).
"""
import os
import sys

import torch  # noqa: F401  (first: see module docstring)
import torch.distributed as dist
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
os.environ.setdefault("OMP_NUM_THREADS", "2")

import _oracle  # noqa: E402
import multigrid_poisson_solver_amd as mg  # noqa: E402


def exchange(a, lo, hi, G, rank, world):
    """ghost rows of a full-size array whose valid rows are [lo, hi)"""
    reqs = []
    t = torch.from_numpy(a)
    if rank + 1 < world:
        reqs.append(dist.isend(t[hi - G:hi].clone(), rank + 1))
        up = torch.empty_like(t[hi:hi + G])
        reqs.append(dist.irecv(up, rank + 1))
    if rank > 0:
        reqs.append(dist.isend(t[lo:lo + G].clone(), rank - 1))
        dn = torch.empty_like(t[lo - G:lo])
        reqs.append(dist.irecv(dn, rank - 1))
    for r in reqs:
        r.wait()
    if rank + 1 < world:
        t[hi:hi + G] = up
    if rank > 0:
        t[lo - G:lo] = dn


def window_array(full, lo, hi, G):
    out = np.full_like(full, np.nan)
    a, b = max(0, lo - G), min(full.shape[0], hi + G)
    out[a:b] = full[a:b]
    return out


def main():
    N, collapse, step, path = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    orc = _oracle.Oracle()
    G = mg.slab_ghost_rows()                                  # halo rows a slab carries
    depth = mg.slab_ghost_depths(N, 8, world, collapse, step)  # ... and how many of them travel, per level
    levels = mg.slab_partition(N, 8, world, collapse)
    sizes = [l[0] for l in levels]
    L = 1.0
    first_collapsed = next(i for i, l in enumerate(levels) if l[1])

    U = [None] * len(levels)
    F = [None] * len(levels)
    own = [l[2][rank] for l in levels]
    F[0] = window_array(orc.getSource(N), own[0][0], own[0][1], G)

    # ---- down through the distributed levels
    for l in range(first_collapsed):
        n, M = sizes[l], sizes[l + 1]
        lo, hi = own[l]
        start = np.full((n, n), np.nan)
        start[max(0, lo - G):min(n, hi + G)] = 0.0           # memset(U, 0), window only
        U[l], _ = orc.doSmoothing(n, L, start, F[l], step)
        assert not np.isnan(U[l][lo:hi]).any(), f"rank {rank}: NaN in owned U rows of level {n}"
        Fc = orc.doRestriction(n, -orc.getResidual(n, L, U[l], F[l]), M)
        table_lo, _w = mg.restriction_table(n, M)
        if l + 1 < first_collapsed:
            clo, chi = own[l + 1]
        else:  # collapse boundary: ownership induced by this level's partition
            rows = [rc for rc in range(1, M - 1) if lo <= table_lo[rc] < hi]
            clo, chi = (rows[0], rows[-1] + 1) if rows else (1, 1)
            if rank == 0:
                clo = 0
            if rank == world - 1:
                chi = M
        assert not np.isnan(Fc[clo:chi]).any(), f"rank {rank}: NaN in owned coarse F rows {M}"
        # the engine exchanges this level's U halo right here, in the same group as the next
        # level's F halo: it is needed when the cycle comes back up through this level
        Ul = np.full((n, n), np.nan)
        Ul[lo:hi] = U[l][lo:hi]
        exchange(Ul, lo, hi, depth[l], rank, world)
        U[l] = Ul
        if l + 1 < first_collapsed:
            F[l + 1] = np.full((M, M), np.nan)
            F[l + 1][clo:chi] = Fc[clo:chi]
            exchange(F[l + 1], clo, chi, depth[l + 1], rank, world)
        else:
            parts = [None] * world
            dist.all_gather_object(parts, (clo, chi, Fc[clo:chi].copy()))
            full = np.full((M, M), np.nan)
            for a, b, rows in parts:
                full[a:b] = rows
            assert not np.isnan(full).any()
            F[l + 1] = full  # every rank holds the whole collapsed level

    # ---- the collapsed part of the V-cycle, replicated on EVERY rank (whole grids, plain oracle
    # operators): no broadcast of the coarse correction is needed afterwards
    lc = first_collapsed
    coarse_U = np.empty((sizes[lc], sizes[lc]))
    if True:
        def vcycle(l, Fl):
            n = sizes[l]
            if l == len(sizes) - 1:
                return orc.doExactSolver(n, L, Fl, 1e-7)
            Ul, _ = orc.doSmoothing(n, L, np.zeros((n, n)), Fl, step)
            Fc = orc.doRestriction(n, -orc.getResidual(n, L, Ul, Fl), sizes[l + 1])
            Uc = vcycle(l + 1, Fc)
            Ul = orc.doGridAddition(n, Ul, orc.doProlongation(sizes[l + 1], Uc, n, fill=0.0))
            Ul, _ = orc.doSmoothing(n, L, Ul, Fl, step)
            return Ul
        coarse_U[:] = vcycle(lc, F[lc])
    U[lc] = coarse_U

    # ---- up through the distributed levels
    for l in range(first_collapsed - 1, -1, -1):
        n, M = sizes[l], sizes[l + 1]
        lo, hi = own[l]
        # NO exchange on the way up: the coarse halo rows this level reads through the prolongation are
        # the ones this rank computed redundantly itself (whole-grid operators on NaN-poisoned windows
        # compute every row whose inputs are there); GHOST is sized so that this reaches far enough
        Uc = U[l + 1]
        Uf = U[l]  # owned rows + the halo exchanged right after this level's descent
        Uf = orc.doGridAddition(n, Uf, orc.doProlongation(M, Uc, n, fill=np.nan))
        U[l], _ = orc.doSmoothing(n, L, Uf, F[l], step)
        assert not np.isnan(U[l][lo:hi]).any(), f"rank {rank}: NaN in owned rows after the way up, level {n}"

    parts = [None] * world
    dist.all_gather_object(parts, (own[0][0], own[0][1], U[0][own[0][0]:own[0][1]].copy()))
    if rank == 0:
        got = np.empty((N, N))
        for a, b, rows in parts:
            got[a:b] = rows
        want = orc.run_cycle_file(path)
        assert want["status"] == 0
        same = np.array_equal((got + 0.0).view(np.uint64), (want["U"] + 0.0).view(np.uint64))
        print("SLAB_EMULATION", "OK" if same else "MISMATCH", world, N, collapse, flush=True)
        if not same:
            sys.exit(3)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
