"""World-size-N gloo worker (CPU): emulates the row-slab V-cycle of mg_slab.cpp with the CPU
oracle as the local operator.  Every rank holds full-size arrays that are NaN wherever the engine's
launches would not have written (a launch leaves exactly the rows the schedule says it updates),
applies the oracle's whole-grid operators, exchanges exactly the ghost rows the schedule lets travel
(over torch.distributed/gloo) and finally the assembled finest U must equal the oracle's
single-domain result bit for bit with no NaN in any row a launch is supposed to produce -- which
proves the partition and the communication-avoiding schedule (mg_slab_partition / mg_slab_schedule,
the product's host code): which rows are recomputed, which travel, how deep the halos are.  TEST INFRASTRUCTURE (uses oracle/).

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


def keep_rows(a, span):
    """what a launch that updates rows [lo, hi) leaves behind: those rows, NaN everywhere else"""
    out = np.full_like(a, np.nan)
    out[span[0]:span[1]] = a[span[0]:span[1]]
    return out


def main():
    N, collapse, step, path = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    ca_mode, ca_pct = int(sys.argv[5]), int(sys.argv[6])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    orc = _oracle.Oracle()
    # the product's own host code: partition + communication-avoiding schedule (mg_slab.cpp:slab_schedule)
    sched = mg.slab_schedule(N, 8, world, collapse, step, ca_mode, ca_pct)
    sizes = [d["N"] for d in sched]
    L = 1.0
    first_collapsed = next(i for i, d in enumerate(sched) if d["collapsed"])

    U = [None] * len(sched)
    F = [None] * len(sched)
    own = [d["own"][rank] if not d["collapsed"] else None for d in sched]
    # getSource is evaluated by every rank on its whole window (owned rows + halo): no exchange of the finest F
    F[0] = window_array(orc.getSource(N), own[0][0], own[0][1], sched[0]["halo"])
    exchanges = 0

    # ---- down through the distributed levels
    for l in range(first_collapsed):
        n, M = sizes[l], sizes[l + 1]
        lo, hi = own[l]
        dext = sched[l]["dext"][rank]
        # memset(U, 0) is folded into the launch: the zero start needs no U halo at all
        Ul, _ = orc.doSmoothing(n, L, np.zeros((n, n)), F[l], step)
        assert not np.isnan(Ul[dext[0]:dext[1]]).any(), f"rank {rank}: NaN in the rows the -1 launch of level {n} updates"
        U[l] = keep_rows(Ul, dext)   # the launch writes exactly these rows of U ...
        if sched[l]["pre"]:          # ... or none at all where the level's `1` launch recomputes the field from F
            U[l] = np.full((n, n), np.nan)
            assert sched[l]["xU"] == 0, "a recomputing level has no U halo to exchange"
        Fc = orc.doRestriction(n, -orc.getResidual(n, L, Ul, F[l]), M)
        table_lo, _w = mg.restriction_table(n, M)
        if l + 1 < first_collapsed:
            clo, chi = own[l + 1]
            fwr = sched[l + 1]["fwr"][rank]
            # ... and the coarse F rows whose lower-left sample lies in them
            rows = [rc for rc in range(1, M - 1) if dext[0] <= table_lo[rc] < dext[1]]
            want = (0 if dext[0] == 0 else rows[0], M if dext[1] == n else rows[-1] + 1)
            assert tuple(fwr) == want, f"rank {rank}: fwr of level {M}: {fwr} vs {want}"
            assert fwr[0] >= clo - sched[l + 1]["halo"] and fwr[1] <= chi + sched[l + 1]["halo"], "written rows leave the window"
            assert not np.isnan(Fc[fwr[0]:fwr[1]]).any(), f"rank {rank}: NaN in the coarse F rows written at level {M}"
            F[l + 1] = keep_rows(Fc, fwr)
        else:  # collapse boundary: ownership induced by this level's partition
            rows = [rc for rc in range(1, M - 1) if lo <= table_lo[rc] < hi]
            clo, chi = (rows[0], rows[-1] + 1) if rows else (1, 1)
            if rank == 0:
                clo = 0
            if rank == world - 1:
                chi = M
            assert not np.isnan(Fc[clo:chi]).any(), f"rank {rank}: NaN in owned coarse F rows {M}"
        # ONE group after the launch: this level's U halo where the schedule lets it travel (needed only when the
        # cycle comes back up), the next level's F halo where it is exchanged instead of recomputed, the all-gather
        if sched[l]["xU"] > 0:
            exchange(U[l], lo, hi, sched[l]["xU"], rank, world)
            exchanges += 1
        if l + 1 < first_collapsed:
            if sched[l + 1]["xF"] > 0:
                exchange(F[l + 1], clo, chi, sched[l + 1]["xF"], rank, world)
                exchanges += 1
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
        ext = sched[l]["ext"][rank]
        # NO exchange on the way up: the coarse halo rows this level reads through the prolongation are the ones
        # this rank computed redundantly itself one node earlier (ext of the coarser level), and this level's own U
        # and F halos are there since the descent (recomputed, or exchanged right after the level's -1 launch)
        Uc = U[l + 1]
        if sched[l]["pre"]:
            # the recomputing `1` launch: the pre-smoothed field once more, from zero, out of the F rows of the window
            U[l], _ = orc.doSmoothing(n, L, np.zeros((n, n)), F[l], sched[l]["pre"])
        Uf = orc.doGridAddition(n, U[l], orc.doProlongation(M, Uc, n, fill=np.nan))
        Ul, _ = orc.doSmoothing(n, L, Uf, F[l], step)
        assert not np.isnan(Ul[ext[0]:ext[1]]).any(), f"rank {rank}: NaN in the rows the 1 launch of level {n} updates"
        U[l] = keep_rows(Ul, ext)

    parts = [None] * world
    dist.all_gather_object(parts, (own[0][0], own[0][1], U[0][own[0][0]:own[0][1]].copy()))
    if rank == 0:
        got = np.empty((N, N))
        for a, b, rows in parts:
            got[a:b] = rows
        want = orc.run_cycle_file(path)
        assert want["status"] == 0
        same = np.array_equal((got + 0.0).view(np.uint64), (want["U"] + 0.0).view(np.uint64))
        print("SLAB_EMULATION", "OK" if same else "MISMATCH", world, N, collapse, f"mode {ca_mode} exchanges {exchanges}", flush=True)
        if not same:
            sys.exit(3)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
