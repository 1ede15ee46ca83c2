"""bench_multi.py -- the --gpus N > 1 leg of bench.py: one process per GPU (launched by
torch.distributed.run, or by bench.py itself), 1-D row slabs, ghost rows over RCCL/xGMI.

Two legs per run, both in the one JSON line rank 0 prints:
  weak   (`value`): the per-GPU share of the finest grid stays at the 8192^2 points the single-GPU metric is
         quoted on, so the grid is N_g x N_g with N_g ~ 8192*sqrt(gpus) (8192, 11520, 16384, 23040 for 1, 2, 4, 8
         GPUs: sizes m * 2^j with m <= 64, so that every level above the coarse-tail kernel keeps an even size);
  strong (`strong_scaling`): the V-cycle at N = 16384^2 (BASELINE.json configs[3]; north_star's >= 6x at 8 GPUs
         is quoted on it) cut into `gpus` slabs -- the one-GPU base of that curve is in bench.py's N = 1 line;
         `strong_scaling_8192`: the headline grid itself (BASELINE.json's metric: "8192^2 fp64, 1/2/4/8 GPUs") cut into
         `gpus` slabs, base = the N = 1 line's `value`; `strong_scaling_32768`: the same one size up.
`value` is the whole-job aggregate: lattice updates of one V(3,3)-cycle over ALL slabs divided by the slowest
rank's time.  --scaling strong swaps which leg is `value`.

torch is plumbing here (process group for the rendezvous and the timing all-reduce); the engine binds to the HIP
runtime torch brought along (one runtime per process) and resolves RCCL lazily from the copy torch has mapped.
"""
import json
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


def run_leg(mg, args, rank, world, N, rehearsal, tmp):
    """One timed leg: W warm-up windows, then exactly K windows back to back between barriers; max over ranks."""
    from bench import level_sizes, vcycle_algorithmic_bytes, vcycle_compulsory_bytes
    nu = args.nu
    sizes = level_sizes(N, args.n_min)
    cyc = os.path.join(tmp, f"Vcycle_{N}.txt")
    mg.write_vcycle_file(cyc, N, args.n_min, nu, 1e-7)
    lups = sum(2 * nu * s * s for s in sizes[:-1])
    collapse_N = int(os.environ.get("MG_COLLAPSE_N", "1024"))
    refine = max(1, getattr(args, "refine", 1)) if args.mixed else 1   # configs[4]: fp32 cycles joined by fp64 residual + correction
    lups *= refine
    plan = mg.SlabPlan(cyc, world, rank if world > 1 else -1, collapse_N, mixed=args.mixed, refinement=refine)
    for _ in range(max(1, args.warmup)):
        r = plan.execute()
        assert r["status"] == 0, r

    plan.want_error(False)  # the analytic error is outside the reference's window (:434-445)
    dist.barrier()
    torch.cuda.synchronize()
    mg.sync()
    mg.profile_begin(min_N=0)   # every launch and every ghost exchange of rank 0 (hipEvent pairs)
    t0 = time.perf_counter()
    for _ in range(args.steps):   # back to back on the engine's streams, no per-step host sync
        plan.enqueue()
    mg.sync()
    torch.cuda.synchronize()
    dist.barrier()
    elapsed = time.perf_counter() - t0
    prof = mg.profile_end()
    r = plan.collect()
    assert r["status"] == 0, r
    plan.want_error(True)
    r = plan.execute()  # untimed: the result's error against the analytic solution
    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
    every = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(every, t)   # each rank's own clock around the same K windows (the barriers bracket all of them)
    per_rank_ms = [float(e.item()) * 1e3 / args.steps for e in every]
    ms_per_step = max(per_rank_ms)   # the contract: MAX over ranks
    plan.close()
    mg.lib().mg_pool_trim()
    if rank != 0:
        return None

    elem = 0.5 if args.mixed else 1.0
    is_x = lambda e: e["name"].startswith("ghost_exchange")
    exchanges = [{"level_N": e["N"], "what": e["name"], "per_step": e["launches"] // max(1, args.steps),
                  "avg_ms": round(e["total_ms"] / max(1, e["launches"]), 4), "bytes_per_neighbour_pair": e["algo_bytes"]}
                 for e in sorted(prof, key=lambda e: -e["N"]) if is_x(e)]
    launches_ms = sum(e["total_ms"] for e in prof if not is_x(e)) / max(1, args.steps)
    exchange_ms = sum(e["total_ms"] for e in prof if is_x(e)) / max(1, args.steps)
    per_level = {}
    for e in prof:
        if not is_x(e):
            per_level[e["N"]] = per_level.get(e["N"], 0.0) + e["total_ms"] / max(1, args.steps)
    kernels = []
    for e in sorted((e for e in prof if e["N"] == N and not is_x(e)), key=lambda e: -e["total_ms"]):
        avg = e["total_ms"] / max(1, e["launches"])
        # compulsory bytes of a slab launch: the node kernel's bytes on this rank's share of the rows
        share = 1.0 / world
        n = float(N) * N * share
        # (noU / pre3: the node pair that neither stores nor re-reads the pre-smoothed field, levels >= 4096)
        if "restrict" in e["name"]:
            cb = (8.0 if "noU" in e["name"] else 16.0) * n + 2.0 * n
        elif "prolong" in e["name"]:
            cb = (16.0 if "pre" in e["name"] else 24.0) * n + 2.0 * n
        else:
            cb = 24.0 * n
        kernels.append({"kernel": e["name"], "N": e["N"], "launches": e["launches"], "avg_ms": round(avg, 4),
                        "compulsory_GBs": round(cb * elem / (avg * 1e-3) / 1e9, 1) if avg > 0 else None,
                        "algorithmic_equiv_GBs": round(e["algo_bytes"] / (avg * 1e-3) / 1e9, 1) if avg > 0 else None})
    roof = None
    if kernels:
        k0 = kernels[0]
        roof = {"bound": "hbm", "kernel": k0["kernel"] + " (rank 0's slab)", "achieved": k0["compulsory_GBs"],
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round((k0["compulsory_GBs"] or 0.0) / HBM_PEAK_GBS, 4),
                "traffic": None, "avg_ms": k0["avg_ms"], "launches": k0["launches"],
                "algorithmic_equiv": {"GBs": k0["algorithmic_equiv_GBs"]}}
    refine_text = f"{refine} fp32 cycles per window joined by the fp64 residual of the fp64 iterate and an fp64 correction, " if refine > 1 else ""
    cb = vcycle_compulsory_bytes(sizes) * elem * refine + (refine - 1) * 40.0 * N * N   # (joint: 8+8+4 B/pt residual, 4+8+8 B/pt correction)
    gbs = cb / (ms_per_step * 1e-3) / 1e9
    return {
        "N": N, "value": round(lups / (ms_per_step * 1e-3) / 1e6, 1), "unit": "MLUPS", "n_gpus": world, "steps": args.steps,
        "ms_per_step": round(ms_per_step, 4),
        "ms_per_step_ranks": {"min": round(min(per_rank_ms), 4), "max": round(max(per_rank_ms), 4),
                              "all": [round(v, 4) for v in per_rank_ms]},
        "workload": f"V({nu},{nu})-cycle N={N}^2 {'fp32 cycle (mixed mode)' if args.mixed else 'fp64'} ({N * N // world} points per GPU), "
                    f"{len(sizes)} levels, {refine_text}{world} row slabs, communication-avoiding schedule (F halos recomputed, the pre-smoothed U of the levels >= 4096 recomputed instead of "
                    f"stored/re-read/exchanged, ONE RCCL group per cycle on a second stream: collapse all-gather + one U halo), levels N<={collapse_N} replicated on every rank",
        "levels": len(sizes), "fine_dof_per_s": round(N * N / (ms_per_step * 1e-3), 1), "mg_error": r["mg_error"],
        "roofline": roof, "kernels": kernels[:6],
        # where rank 0's time went (live hipEvent pairs): its kernel launches, its ghost exchanges (which
        # include waiting for the neighbours and run beside the interior launches), the rest (gaps)
        "rank0_ms_per_step": {"kernels": round(launches_ms, 4), "ghost_exchanges": round(exchange_ms, 4),
                              "kernels_per_level": {str(k): round(v, 4) for k, v in sorted(per_level.items(), reverse=True)}},
        "ghost_exchanges": exchanges,
        "cycle_roofline": {"compulsory_bytes": cb, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                           "frac": round(gbs / (HBM_PEAK_GBS * world), 4),
                           "algorithmic_equiv": {"bytes": vcycle_algorithmic_bytes(sizes, nu, nu)}},
    }


def run(args, rank, world, local_rank):
    # a rank that waits for ever on a neighbour (a lost peer, a wedged collective) ends with its stacks on stderr and a
    # non-zero exit instead of holding the node until the driver's own limit
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("MG_BENCH_DEADLINE", "900")), exit=True)
    rehearsal = os.environ.get("MG_BENCH_TRANSPORT", "rccl") == "host"
    if rehearsal:
        local_rank = 0  # every rank on the one GPU
    torch.cuda.set_device(local_rank)
    single_process = "WORLD_SIZE" not in os.environ  # --force-slab on one rank without torchrun
    if single_process:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    if rehearsal or single_process:
        dist.init_process_group(backend="gloo")
        rehearsal_wire = True
    else:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        rehearsal_wire = False
    import multigrid_poisson_solver_amd as mg
    from bench import STRONG_N, grid_for

    mg.init(local_rank)
    mg.set_smoother("stream")
    wire = {"transport": "none (one rank)", "nranks": 1, "lib": "", "selftest": None}
    if world > 1:
        if rehearsal:
            host_transport(mg, rank, world)
        else:
            uid = [mg.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            mg.comm_init(rank, world, uid[0])
            # sanity gate before anything is timed: the calls the slab driver makes (grouped send/recv on a second
            # stream, all-gather over all ranks) must deliver their bytes
            if mg.lib().mg_comm_selftest(1 << 16) != 0:
                raise SystemExit(f"rank {rank}: RCCL self-test failed: {mg.lib().mg_last_error_string()}")
        # what the record needs to certify itself: the communicator's own rank count (not the environment's), the shared
        # object the nccl* entry points were resolved from, the self-test's verdict -- from EVERY rank
        mine = {"rank": mg.lib().mg_comm_rank(), "nranks": mg.lib().mg_comm_size(), "lib": (mg.lib().mg_comm_library() or b"").decode(),
                "device": local_rank}
        seen = [None] * world
        dist.all_gather_object(seen, mine)
        wire = {"transport": "host-staged over gloo (rehearsal)" if rehearsal else "rccl",
                "nranks": mine["nranks"], "lib": mine["lib"], "selftest": None if rehearsal else 0,
                "ranks_agree": all(x["nranks"] == world and x["lib"] == mine["lib"] for x in seen) and
                               sorted(x["rank"] for x in seen) == list(range(world)),
                "devices": [x["device"] for x in seen]}

    tmp = tempfile.mkdtemp(prefix=f"mgbench_r{rank}_")
    N_weak = grid_for(world, args.n, args.mixed) if args.n == 8192 else args.n
    # The strong-scaling legs run first: they are legs of the line AND the sustained load after which the weak leg -- the
    # line's `value` -- starts at the clocks a long run holds (bench.py does the same on one GPU).
    keys = ("N", "n_gpus", "value", "unit", "ms_per_step", "steps", "workload", "mg_error", "rank0_ms_per_step", "ghost_exchanges",
            "cycle_roofline", "ms_per_step_ranks")
    strong = None
    if not args.no_strong and N_weak != STRONG_N:
        strong = run_leg(mg, args, rank, world, STRONG_N, rehearsal_wire, tmp)
    large = None
    if not args.no_strong and not getattr(args, "no_large", False) and not args.mixed:
        from bench import LARGE_N
        import copy
        largs = copy.copy(args)
        largs.steps, largs.warmup = min(args.steps, 10), min(max(1, args.warmup), 10)
        large = run_leg(mg, largs, rank, world, LARGE_N, rehearsal_wire, tmp)
    # BASELINE.json's metric names 8192^2 at 1/2/4/8 GPUs: the headline grid itself cut into `world` slabs (1024 rows per
    # slab at 8 GPUs: three distributed levels, the hierarchy collapses at 1024) -- a strong leg whose one-GPU base is the
    # N = 1 line's `value`
    head_grid = None
    if not args.no_strong and world > 1 and N_weak != 8192 and not args.mixed:
        head_grid = run_leg(mg, args, rank, world, 8192, rehearsal_wire, tmp)
    weak = run_leg(mg, args, rank, world, N_weak, rehearsal_wire, tmp)
    if not args.no_strong and N_weak == STRONG_N:
        strong = weak  # 4 GPUs: the weak-scaling grid IS 16384^2

    if rank == 0:
        head, other, name = (weak, strong, "strong_scaling")
        if args.scaling == "strong" and strong:
            head, other, name = (strong, weak, "weak_scaling")
        out = {
            "metric": "vcycle_mlups", "value": head["value"], "unit": "MLUPS", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True,
            "scaling": "strong" if head is strong and args.scaling == "strong" else "weak", "vs_baseline": None,
            "dtype": "f32" if args.mixed else "f64", "data": "synthetic",
            "config": {"workload": head["workload"], "N": head["N"], "levels": head["levels"], "parallelism": f"slab{world}"},
            "fine_dof_per_s": head["fine_dof_per_s"], "mg_error": head["mg_error"], "roofline": head["roofline"],
            "kernels": head["kernels"], "rank0_ms_per_step": head["rank0_ms_per_step"], "ghost_exchanges": head["ghost_exchanges"],
            "cycle_roofline": head["cycle_roofline"], "ms_per_step_ranks": head["ms_per_step_ranks"], "rccl": wire,
            **({"transport": "host-staged over gloo, all ranks on ONE GPU: a plumbing rehearsal, not a measurement"}
               if rehearsal and world > 1 else {}),
        }
        if other:
            out[name] = {k: other[k] for k in keys}
            out[name]["scaling"] = "strong" if name == "strong_scaling" else "weak"
        if large:
            out["strong_scaling_32768"] = {k: large[k] for k in keys}
            out["strong_scaling_32768"]["scaling"] = "strong"
        if head_grid:
            out["strong_scaling_8192"] = {k: head_grid[k] for k in keys}
            out["strong_scaling_8192"]["scaling"] = "strong"
        print(json.dumps(out), flush=True)
    if world > 1:
        mg.lib().mg_comm_finalize()
    mg.finalize()
    dist.barrier()
    dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()
