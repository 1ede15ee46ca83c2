#!/usr/bin/env python3
"""bench.py -- V-cycle MLUPS + achieved HBM GB/s vs roofline (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: ONE run of the
reference program's timed window (src/MG_solver_CPU.cpp:156..429) over a generated
V(3,3) cycle-structure file -- level push/pop, smoothing, residual, restriction, coarse
Gauss-Seidel solve, prolongation, correction -- from the state right after getSource,
inputs resident in HBM.  At --gpus 1 the workload is the configuration the metric is
quoted on: N = 8192^2 fp64.  Prints ONE JSON line (see the driver's contract).
"""
import argparse
import json
import math
import os

os.environ.setdefault("OMP_WAIT_POLICY", "passive")  # before libgomp loads (cpu_baseline leg)
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def level_sizes(N, N_min):
    out = []
    while N >= N_min:
        out.append(N)
        N //= 2
    return out


def grid_for(world, base=8192, mixed=False):
    """Weak scaling: ~base^2 points per GPU.  Among the sizes m * 2^j with m <= 64 (every level above the
    coarse-tail kernel's N <= 64 is then even, so every node runs in its fused one-launch form) the one nearest
    to base * sqrt(world): 8192, 11520, 16384, 23040 for 1, 2, 4, 8 GPUs."""
    target = base * math.sqrt(world)
    best = None
    for m in range(33, 65):
        j = max(0, round(math.log2(target / m)))
        for jj in (j - 1, j, j + 1):
            if jj < 1:
                continue
            n = m * 2 ** jj
            if best is None or abs(n - target) < abs(best - target):
                best = n
    return int(best)


def vcycle_algorithmic_bytes(sizes, nu1, nu2):
    """B = sum_{l<Lc} [(8 + 24 nu1 + 24 + 8 + 16 + 24 nu2) n_l + 16 n_{l+1}]."""
    total = 0.0
    for a, b in zip(sizes[:-1], sizes[1:]):
        total += (8 + 24 * nu1 + 24 + 8 + 16 + 24 * nu2) * a * a + 16.0 * b * b
    return total


def measured_traffic(kernel_family):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r01_traffic.json, produced by scripts/profile.sh on the same command)."""
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    if not os.path.exists(path):
        return None
    data = json.load(open(path)).get("kernels", {})
    import re
    m = re.match(r"jacobi_stream<(\d)(.*)>", kernel_family)
    if not m:
        return None
    steps, rest = m.group(1), m.group(2)
    mode = "2" if "prolong" in rest else ("1" if "zero" in rest else "0")
    restrict = "true" if "restrict" in rest else "false"
    # the finest-level launch is the variant that moved the most bytes (the same template with
    # another prefetch depth serves the small levels)
    hits = [v["total_bytes"] for name, v in data.items()
            if name.split("::")[-1].startswith(f"k_jacobi_stream<{steps}, 2, {mode}, {restrict}") and not name.startswith("f32::")]
    return max(hits) if hits else None


def compulsory_bytes(kernel_family, N):
    """HBM bytes one launch of a temporally blocked node kernel cannot avoid (kernel
    table): every input array read once, every output written once.  `-1` node: F in, U out,
    coarse F out (U is zero-filled in registers); `1` node: U, F, coarse U in, U out; plain
    S-sweep launch: U, F in, U out.  2 B/pt of table/halo overhead are not counted."""
    n = float(N) * N
    if "restrict" in kernel_family:
        return 16.0 * n + 8.0 * (N // 2) ** 2
    if "prolong" in kernel_family:
        return 24.0 * n + 8.0 * (N // 2) ** 2
    if "jacobi_stream" in kernel_family or "jacobi_pair" in kernel_family:
        return (16.0 if "zero" in kernel_family else 24.0) * n
    return None


def cpu_baseline(cycle_path, lups, threads=None):
    """The reference's own operators (oracle/_ref/libmgref.so, built from /root/reference by
    oracle/Makefile) -- or, when that build is absent, the oracle's restatement -- timed on
    this host's cores over ONE run of the same cycle file, in the reference's own window.
    TEST/BASELINE infrastructure: reported beside the GPU number, never part of it."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle
    orc = _oracle.Oracle()
    # a 1-GPU job owns 16 host cores on the GPU box, whatever os.cpu_count() says
    cores = threads or min(16, os.cpu_count() or 1)
    orc.set_threads(cores)
    ops, kind = None, "port"
    if _oracle.have_reference():
        ops, kind = _oracle.Reference(), "reference"
        ops.set_threads(cores)
    res = orc.run_cycle_file(cycle_path, ops=ops, want_report=False)
    if res["status"] != 0:
        return None
    return {"value": round(lups / (res["time_ms"] * 1e-3) / 1e6, 3), "unit": "MLUPS", "cores": cores, "kind": kind,
            "time_ms": round(res["time_ms"], 2), "mg_error": res["mg_error"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=int(os.environ.get("MG_BENCH_N", "8192")),
                    help="finest grid size N (N x N points); under torchrun use env MG_BENCH_N (its parser eats --n)")
    ap.add_argument("--n-min", type=int, default=8)
    ap.add_argument("--nu", type=int, default=3, help="smoothing steps per node (V(nu,nu))")
    ap.add_argument("--cycle", choices=["V", "W"], default="V")
    ap.add_argument("--mode", choices=["eager", "graph", "unfused"], default="eager")
    ap.add_argument("--smoother", choices=["stream", "simple"], default=os.environ.get("MG_SMOOTHER", "stream"))
    ap.add_argument("--cpu-n", type=int, default=None, help="grid size of the CPU baseline sample (default: --n)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--mixed", action="store_true",
                    help="mixed-precision mode (fp32 cycle, fp64 source/result; NOT the headline metric, which is fp64)")
    ap.add_argument("--force-slab", action="store_true",
                    help="run the row-slab/RCCL leg even with one rank (plumbing rehearsal on a 1-GPU box)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 or args.force_slab:
        from multigrid_poisson_solver_amd import build as mg_build
        mg_build.ensure_built()
        import bench_multi  # row-slab path (one process per GPU, RCCL ghost rows)
        return bench_multi.run(args, rank, world, local_rank)

    from multigrid_poisson_solver_amd import build as mg_build
    mg_build.ensure_built()   # no-op when the in-tree library is there (the normal case)
    import multigrid_poisson_solver_amd as mg
    mg.init(local_rank)
    mg.set_smoother(args.smoother)

    N, nu = args.n, args.nu
    sizes = level_sizes(N, args.n_min)
    tmp = tempfile.mkdtemp(prefix="mgbench_")
    cyc = os.path.join(tmp, f"{args.cycle}cycle_{N}.txt")
    if args.cycle == "V":
        mg.write_vcycle_file(cyc, N, args.n_min, nu, 1e-7)
        visits = [1] * len(sizes)
    else:
        mg.write_wcycle_file(cyc, N, args.n_min, nu, 1e-7)
        # level l >= 1 is smoothed by 2^l '-1' nodes and 2^l '1' nodes (shipped src/Wcycle.txt recursion)
        visits = [1] + [2 ** l for l in range(1, len(sizes))]
    # lattice updates of one step: (nu1+nu2) * sum over smoothed levels of n_l (x visits for W)
    lups = sum(2 * nu * v * s * s for s, v in zip(sizes[:-1], visits[:-1]))
    algo_bytes = vcycle_algorithmic_bytes(sizes, nu, nu) if args.cycle == "V" else None

    plan = mg.CyclePlan(cyc, fused=(args.mode != "unfused"), graph=(args.mode == "graph"), report=False, error=False,
                        mixed=args.mixed)
    first = None
    for _ in range(max(args.warmup, 2 if args.mode == "graph" else 0)):
        first = plan.execute()
        assert first["status"] == 0, first

    # ---- timed region: exactly K steps, bracketed by synchronisation on both sides ----
    mg.sync()
    mg.profile_begin(min_N=N)          # hipEvent pairs around the finest-level launches
    t0 = time.perf_counter()
    dev_ms = 0.0
    # the K windows are enqueued back to back on the engine's stream (no per-step host sync: a
    # fixed-step cycle file needs none) and the region ends with one synchronisation
    for _ in range(args.steps):
        plan.enqueue()
    mg.sync()
    t1 = time.perf_counter()
    r = plan.collect()
    dev_ms = r["device_ms"] * args.steps
    prof = mg.profile_end()
    assert r["status"] == 0
    ms_per_step = (t1 - t0) * 1e3 / args.steps
    mlups = lups / (ms_per_step * 1e-3) / 1e6

    # dominant kernel = the launch family with the largest total time on the finest grid
    roof = None
    kernels = []
    for e in sorted(prof, key=lambda e: -e["total_ms"]):
        avg = e["total_ms"] / max(1, e["launches"])
        gbs = e["algo_bytes"] / (avg * 1e-3) / 1e9 if avg > 0 else 0.0
        kernels.append({"kernel": e["name"], "N": e["N"], "launches": e["launches"], "avg_ms": round(avg, 4),
                        "algo_GBs": round(gbs, 1)})
    if kernels:
        k0 = kernels[0]
        roof = {"bound": "hbm", "kernel": k0["kernel"], "achieved": k0["algo_GBs"], "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(k0["algo_GBs"] / HBM_PEAK_GBS, 4),
                "traffic": measured_traffic(k0["kernel"]) if N == 8192 and not args.mixed else None,
                "avg_ms": k0["avg_ms"], "launches": k0["launches"]}
        # `achieved` prices the launch at SURVEY section 8d's 24 B per lattice update, which the
        # temporally blocked kernel undercuts (S sweeps per pass over HBM): frac > 1 is that
        # saving, not a bandwidth.  What the launch really has to move, and how fast it moves it:
        cb = compulsory_bytes(k0["kernel"], N)
        if cb and args.mixed:
            cb /= 2
        if cb:
            g = cb / (k0["avg_ms"] * 1e-3) / 1e9
            roof["hbm"] = {"compulsory_bytes": cb, "achieved": round(g, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(g / HBM_PEAK_GBS, 4)}

    out = {
        "metric": "vcycle_mlups", "value": round(mlups, 1), "unit": "MLUPS", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32" if args.mixed else "f64", "data": "synthetic",
        "config": {"workload": f"{args.cycle}({nu},{nu})-cycle N={N}^2 {'fp32 cycle (mixed mode)' if args.mixed else 'fp64'}, {len(sizes)} levels to N={sizes[-1]}, "
                               f"red-black GS(1e-7) coarse solve, cycle-file driver ({args.mode}, {args.smoother} smoother)",
                   "N": N, "levels": len(sizes), "cycle_file": os.path.basename(cyc)},
        "device_ms_per_step": round(dev_ms / args.steps, 4),
        "fine_dof_per_s": round(N * N / (ms_per_step * 1e-3), 1),
        "mg_error": plan.analytic_error(r),
        "roofline": roof,
        "kernels": kernels[:8],
    }
    if algo_bytes:
        gbs = algo_bytes / (ms_per_step * 1e-3) / 1e9
        out["cycle_roofline"] = {"algorithmic_bytes": algo_bytes, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)}
    plan.close()

    # north_star's own yardstick, measured beside the cycle: ONE fine-level Jacobi sweep per
    # launch (doSmoothing with step = 1, 24 algorithmic bytes per point), both smoothers
    try:
        Ua, Ub, Ff = mg.DeviceGrid.uniform(N, 3), mg.DeviceGrid(N), mg.DeviceGrid.uniform(N, 7)
        single = {}
        for sm in ("stream", "simple"):
            mg.set_smoother(sm)
            for _ in range(3):
                mg.smooth_pp(N, 1.0, Ua, Ub, Ff, 1)
            mg.sync()
            mg.profile_begin(min_N=N)
            for _ in range(10):
                mg.smooth_pp(N, 1.0, Ua, Ub, Ff, 1)
            e = mg.profile_end()[0]
            avg = e["total_ms"] / e["launches"]
            gbs = 24.0 * N * N / (avg * 1e-3) / 1e9
            single[sm] = {"kernel": e["name"], "avg_ms": round(avg, 4), "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                          "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)}
        out["single_sweep_roofline"] = single
        mg.set_smoother(args.smoother)
        # the box's own device-to-device copy rate: the practical ceiling SURVEY.md 8d asks for
        mg.lib().mg_copy(Ub.ptr, Ua.ptr, Ua.size)
        mg.sync()
        tc = time.perf_counter()
        for _ in range(10):
            mg.lib().mg_copy(Ub.ptr, Ua.ptr, Ua.size)
        mg.sync()
        tc = (time.perf_counter() - tc) / 10
        out["device_copy_GBs"] = round(16.0 * N * N / tc / 1e9, 1)
        for g in (Ua, Ub, Ff):
            g.free()
    except mg.MGError as exc:  # never let the side measurement take the bench line down
        out["single_sweep_roofline"] = {"error": str(exc)}

    if not args.no_cpu:
        cpu_n = args.cpu_n or N
        cpu_cyc = cyc
        cpu_lups = lups
        if cpu_n != N or args.cycle != "V":
            cpu_cyc = os.path.join(tmp, f"cpu_Vcycle_{cpu_n}.txt")
            mg.write_vcycle_file(cpu_cyc, cpu_n, args.n_min, nu, 1e-7)
            cs = level_sizes(cpu_n, args.n_min)
            cpu_lups = sum(2 * nu * s * s for s in cs[:-1])
        base = cpu_baseline(cpu_cyc, cpu_lups)
        if base:
            base["sample"] = (f"1 V({nu},{nu})-cycle at N={cpu_n}^2 (the reference's timed window, "
                              f"{'reference operators oracle/_ref' if base['kind'] == 'reference' else 'oracle restatement'}"
                              f", -O2, {base['cores']} OpenMP threads)")
            out["cpu_baseline"] = base
    mg.finalize()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
