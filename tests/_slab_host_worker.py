"""One rank of the REAL row-slab driver (mg_slab_* in rank mode) with the host-staged transport
(mg_comm_init_host) over torch.distributed/gloo: several of these processes share one GPU, which
RCCL does not allow, so this is how the rank-mode exchange code (peers, counts, grouping, the
collapse all-gather, the error all-gather) is verified against the oracle on a 1-GPU box.
torch is imported before the engine library (same HIP runtime)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    cycle_path, want_path, collapse = sys.argv[1], sys.argv[2], int(sys.argv[3])
    mixed = len(sys.argv) > 4 and sys.argv[4].startswith("mixed")
    cycles = int(sys.argv[4][5:]) if mixed and len(sys.argv[4]) > 5 else 1   # "mixed3": 3 refinement cycles
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    import multigrid_poisson_solver_amd as mg
    mg.init(0)

    def exchange(ops):
        p2p = [dist.P2POp(dist.isend if is_send else dist.irecv, torch.from_numpy(buf), peer) for is_send, peer, buf in ops]
        for req in dist.batch_isend_irecv(p2p):
            req.wait()

    def allgather(send, recv):
        parts = [torch.empty(send.size, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(parts, torch.from_numpy(send.copy()))
        for r in range(world):
            recv[r, :] = parts[r].numpy()

    mg.comm_init_host(rank, world, exchange, allgather)
    want = np.load(want_path)
    N = int(want["N"])
    plan = mg.SlabPlan(cycle_path, world, rank, collapse, mixed=mixed, refinement=cycles)
    for run in range(2):  # the second window re-runs on the state the first one left behind
        res = plan.execute()
        assert res["status"] == 0, res
        lo, hi = mg.slab_partition(N, 8, world, collapse)[0][2][rank]
        U = plan.gather_U(N)
        got, exp = U[lo:hi] + 0.0, want["U"][lo:hi] + 0.0
        assert np.array_equal(got.view(np.uint64), exp.view(np.uint64)), f"rank {rank}: owned rows {lo}:{hi} differ (run {run})"
        recs = [(int(a), int(b)) for a, b, _c, _d in res["records"]]
        assert recs == [(int(a), int(b)) for a, b in want["rec_nodes"]], (recs, want["rec_nodes"])
        errs = np.array([d for _a, _b, _c, d in res["records"]])
        assert np.allclose(errs, want["rec_errors"], rtol=1e-12, atol=0), (errs, want["rec_errors"])
        assert abs(res["mg_error"] - float(want["mg_error"])) <= 1e-10 * float(want["mg_error"])
    plan.close()
    dist.barrier()
    mg.lib().mg_comm_finalize()
    mg.finalize()
    if rank == 0:
        print(f"SLAB_HOST_TRANSPORT OK {world} {N} {collapse}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
