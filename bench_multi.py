"""bench_multi.py -- the --gpus N > 1 leg of bench.py: one process per GPU (launched by
torch.distributed.run), 1-D row slabs, ghost rows over RCCL/xGMI.

Weak scaling: the per-GPU share of the finest grid stays at the 8192^2 points the
single-GPU metric is quoted on, so the grid is N_g x N_g with N_g ~ 8192*sqrt(gpus)
(8192, 11520, 16384, 23040 for 1, 2, 4, 8 GPUs: sizes m * 2^j with m <= 64, so that every level
above the coarse-tail kernel keeps an even size).  `value` is the whole-job aggregate: lattice updates of one
V(3,3)-cycle over ALL slabs divided by the slowest rank's time.

torch is imported BEFORE the engine on purpose: libmgpoisson.so then binds to the HIP and RCCL
runtimes torch already loaded (same sonames) instead of bringing in a second copy.
"""
import json
import math
import os
import tempfile
import time

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0


def host_transport(mg, rank, world):
    """Rehearsal wire (MG_BENCH_TRANSPORT=host): ghost rows staged through the host and carried by
    gloo, so that several ranks can share the one GPU of a test box.  Never the measured path."""
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


def run(args, rank, world, local_rank):
    rehearsal = os.environ.get("MG_BENCH_TRANSPORT", "rccl") == "host"
    if rehearsal:
        local_rank = 0  # every rank on the one GPU
    torch.cuda.set_device(local_rank)
    if rehearsal:
        dist.init_process_group(backend="gloo")
    else:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    import multigrid_poisson_solver_amd as mg
    from bench import grid_for, level_sizes, vcycle_algorithmic_bytes

    mg.init(local_rank)
    mg.set_smoother("stream")
    if rehearsal:
        host_transport(mg, rank, world)
    else:
        uid = [mg.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        mg.comm_init(rank, world, uid[0])

    N = grid_for(world, args.n, args.mixed) if args.n == 8192 else args.n
    nu = args.nu
    sizes = level_sizes(N, args.n_min)
    tmp = tempfile.mkdtemp(prefix=f"mgbench_r{rank}_")
    cyc = os.path.join(tmp, f"Vcycle_{N}.txt")
    mg.write_vcycle_file(cyc, N, args.n_min, nu, 1e-7)
    lups = sum(2 * nu * s * s for s in sizes[:-1])
    algo_bytes = vcycle_algorithmic_bytes(sizes, nu, nu)

    collapse_N = int(os.environ.get("MG_COLLAPSE_N", "1024"))
    plan = mg.SlabPlan(cyc, world, rank, collapse_N, mixed=args.mixed)
    for _ in range(max(1, args.warmup)):
        r = plan.execute()
        assert r["status"] == 0, r

    plan.want_error(False)  # the analytic error is outside the reference's window (:434-445)
    dist.barrier()
    torch.cuda.synchronize()
    mg.profile_begin(min_N=0)   # every launch and every ghost exchange of rank 0 (hipEvent pairs)
    t0 = time.perf_counter()
    for _ in range(args.steps):   # back to back on the engine's stream, no per-step host sync
        plan.enqueue()
    mg.sync()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof = mg.profile_end()
    r = plan.collect()
    assert r["status"] == 0, r
    plan.want_error(True)
    r = plan.execute()  # untimed: the result's error against the analytic solution
    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms_per_step = elapsed * 1e3 / args.steps

    if rank == 0:
        kernels = []
        exchanges = [{"level_N": e["N"], "what": e["name"], "per_step": e["launches"] // max(1, args.steps),
                      "avg_ms": round(e["total_ms"] / max(1, e["launches"]), 4), "bytes_per_neighbour_pair": e["algo_bytes"]}
                     for e in sorted(prof, key=lambda e: -e["N"]) if e["name"].startswith("ghost_exchange")]
        launches_ms = sum(e["total_ms"] for e in prof if not e["name"].startswith("ghost_exchange")) / max(1, args.steps)
        exchange_ms = sum(e["total_ms"] for e in prof if e["name"].startswith("ghost_exchange")) / max(1, args.steps)
        prof = [e for e in prof if e["N"] == N and not e["name"].startswith("ghost_exchange")]
        for e in sorted(prof, key=lambda e: -e["total_ms"]):
            avg = e["total_ms"] / max(1, e["launches"])
            gbs = e["algo_bytes"] / (avg * 1e-3) / 1e9 if avg > 0 else 0.0
            kernels.append({"kernel": e["name"], "N": e["N"], "launches": e["launches"], "avg_ms": round(avg, 4),
                            "algo_GBs": round(gbs, 1)})
        roof = None
        if kernels:
            k0 = kernels[0]
            roof = {"bound": "hbm", "kernel": k0["kernel"] + " (rank 0's slab)", "achieved": k0["algo_GBs"],
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(k0["algo_GBs"] / HBM_PEAK_GBS, 4),
                    "traffic": None, "avg_ms": k0["avg_ms"], "launches": k0["launches"]}
        gbs = algo_bytes / (ms_per_step * 1e-3) / 1e9
        out = {
            "metric": "vcycle_mlups", "value": round(lups / (ms_per_step * 1e-3) / 1e6, 1), "unit": "MLUPS",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.mixed else "f64", "data": "synthetic",
            "config": {"workload": f"V({nu},{nu})-cycle N={N}^2 {'fp32 cycle (mixed mode)' if args.mixed else 'fp64'} ({N * N // world} points per GPU), "
                                   f"{len(sizes)} levels, {world} row slabs, ghost rows over RCCL, levels N<={collapse_N} replicated on every rank",
                       "N": N, "levels": len(sizes), "parallelism": f"slab{world}"},
            "fine_dof_per_s": round(N * N / (ms_per_step * 1e-3), 1),
            "mg_error": r["mg_error"],
            "roofline": roof, "kernels": kernels[:6],
            # where rank 0's time went (live hipEvent pairs): its kernel launches, its ghost exchanges (which
            # include waiting for the neighbours), the rest (gaps)
            "rank0_ms_per_step": {"kernels": round(launches_ms, 4), "ghost_exchanges": round(exchange_ms, 4)},
            "ghost_exchanges": exchanges,
            **({"transport": "host-staged over gloo, all ranks on ONE GPU: a plumbing rehearsal, not a measurement"}
               if rehearsal else {}),
            "cycle_roofline": {"algorithmic_bytes": algo_bytes, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS * world,
                               "unit": "GB/s", "frac": round(gbs / (HBM_PEAK_GBS * world), 4)},
        }
        print(json.dumps(out), flush=True)
    plan.close()
    mg.lib().mg_comm_finalize()
    mg.finalize()
    dist.barrier()
    dist.destroy_process_group()
