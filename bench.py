#!/usr/bin/env python3
"""bench.py -- V-cycle MLUPS + achieved HBM GB/s vs roofline (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: ONE run of the
reference program's timed window (src/MG_solver_CPU.cpp:156..429) over a generated
V(3,3) cycle-structure file -- level push/pop, smoothing, residual, restriction, coarse
Gauss-Seidel solve, prolongation, correction -- from the state right after getSource,
inputs resident in HBM.  At --gpus 1 the workload is the configuration the metric is
quoted on: N = 8192^2 fp64.  Prints ONE JSON line (see the driver's contract).

--gpus N > 1 without a torchrun environment: this process starts the N ranks itself (a child
`python -m torch.distributed.run`, before anything here touches the GPU) and relays rank 0's line.
Every line carries two legs: `value` = weak scaling (8192^2 points per GPU, the headline) and
`strong_scaling` = the V-cycle at N = 16384^2 on the same GPUs (BASELINE.json configs[3]; the
>= 6x target of north_star is quoted there), at N = 1 as the one-GPU base of that curve.
"""
import argparse
import hashlib
import json
import math
import os

os.environ.setdefault("OMP_WAIT_POLICY", "passive")  # before libgomp loads (cpu_baseline leg)
import statistics
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
STRONG_N = 16384       # BASELINE.json configs[3]: V-cycle N=16384^2, row slabs
LARGE_N = 32768        # the same hierarchy one level up (24 GiB of level-0 arrays in fp64): where a rank of an 8-GPU run has
                       # enough work for the latency chain of a cycle to stop capping the strong-scaling ratio


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
    """B = sum_{l<Lc} [(8 + 24 nu1 + 24 + 8 + 16 + 24 nu2) n_l + 16 n_{l+1}]
    (operator by operator: every sweep a pass of its own)."""
    total = 0.0
    for a, b in zip(sizes[:-1], sizes[1:]):
        total += (8 + 24 * nu1 + 24 + 8 + 16 + 24 * nu2) * a * a + 16.0 * b * b
    return total


RECOMPUTE_MIN_N = int(os.environ.get("MG_RECOMPUTE_MIN_N", "4096"))  # the library's default (mg_abi.cpp: recompute_available)


BATCH_RECOMPUTE_MIN_N = int(os.environ.get("MG_RECOMPUTE_MIN_N", os.environ.get("MG_BATCH_RECOMPUTE_MIN_N", "1024")))  # batched schedules (mg_cycle.cpp: build_schedule)


def vcycle_compulsory_bytes(sizes, recompute=True, visits=None, min_n=None):
    """HBM bytes one V-cycle of the FUSED driver cannot avoid: per level one `-1` launch (F in, U out, coarse F
    out: 16 n + 8 m) and one `1` launch (U, F, coarse U in, U out: 24 n + 8 m); from N = 4096 on the pair neither
    writes nor re-reads the pre-smoothed U (the `1` node recomputes it): 8 n + 8 m and 16 n + 8 m.  The coarse tail
    (N <= 64) lives in LDS and is not counted."""
    total = 0.0
    for l, (a, b) in enumerate(zip(sizes[:-1], sizes[1:])):
        if a <= 64:
            break
        # (visits: W-cycle, every visit of a level pair moves its bytes again)
        total += (visits[l] if visits else 1) * ((24.0 if recompute and a >= (min_n or RECOMPUTE_MIN_N) else 40.0) * a * a + 16.0 * b * b)
    return total


def lib_sha():
    import multigrid_poisson_solver_amd as mg
    h = hashlib.sha256()
    with open(mg.LIB_PATH, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()[:16]


def measured_traffic(kernel_family):
    """HBM bytes per launch of the dominant kernel from a committed rocprofv3 PMC run (profiles/rNN_traffic.json,
    scripts/profile.sh: separate FETCH_SIZE / WRITE_SIZE passes over this very command, FETCH x2 on gfx950).  The
    counters cannot be collected inside a timed bench run, so the record is tied to the library it was measured
    on: it is reported only when the sha of the loaded libmgpoisson.so equals the profiled one, else null."""
    import glob
    import re
    m = re.match(r"jacobi_stream<(\d)(.*)>", kernel_family)
    if not m:
        return None
    steps, rest = m.group(1), m.group(2)
    mode = "2" if "prolong" in rest else ("1" if "zero" in rest else "0")
    restrict = "true" if "restrict" in rest else "false"
    sha = lib_sha()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        rec = json.load(open(path))
        if rec.get("lib_sha") != sha:
            continue
        # the finest-level launch is the variant that moved the most bytes (the same template serves the small levels)
        hits = [v["total_bytes"] for name, v in rec.get("kernels", {}).items()
                if name.split("::")[-1].startswith(f"k_jacobi_stream<{steps}, 2, {mode}, {restrict}") and not name.startswith("f32::")]
        if hits:
            return {"bytes": max(hits), "source": os.path.relpath(path, ROOT) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)",
                    "lib_sha": sha}
    return None


def compulsory_bytes(kernel_family, N):
    """HBM bytes one launch of a temporally blocked node kernel cannot avoid (kernel
    table): every input array read once, every output written once.  `-1` node: F in, U out,
    coarse F out (U is zero-filled in registers); `1` node: U, F, coarse U in, U out; plain
    S-sweep launch: U, F in, U out.  2 B/pt of table/halo overhead are not counted."""
    n = float(N) * N
    if "widen" in kernel_family:   # fp32 `1` node that stores its result in fp64: counted in fp64-equivalent units (x 0.5 later)
        # F, coarse U in (fp32), the result out in fp64 (twice an fp32 array); the pre-smoothed U in unless it is recomputed
        return (8.0 if "pre" in kernel_family else 16.0) * n + 8.0 * (N // 2) ** 2 + 16.0 * n
    if "restrict" in kernel_family:   # noU: the smoothed field is not stored (its `1` node recomputes it)
        return (8.0 if "noU" in kernel_family else 16.0) * n + 8.0 * (N // 2) ** 2
    if "prolong" in kernel_family:    # pre3: the pre-smoothed field is recomputed (3 sweeps from zero), not read
        return (16.0 if "pre" in kernel_family else 24.0) * n + 8.0 * (N // 2) ** 2
    if any(f in kernel_family for f in ("jacobi_stream", "jacobi_pair", "jacobi_tile", "slab_stream", "slab_tile")):
        return (16.0 if "zero" in kernel_family else 24.0) * n
    return None


def level_visits(sizes, cycle):
    """How often a level's `-1` / `1` node pair runs in one cycle: once per level in a V-cycle; in the W-cycle of the shipped
    src/Wcycle.txt recursion level l >= 1 runs 2^l times."""
    return [1] * len(sizes) if cycle == "V" else [1] + [2 ** l for l in range(1, len(sizes))]


def cpu_baseline(tmp, N, n_min, nu, write_vcycle_file, cycle="V"):
    """The reference's own operators (oracle/_ref/libmgref*.so, built from /root/reference by oracle/Makefile)
    -- or, when that build is absent, the oracle's restatement -- timed on this host's cores in the reference's
    own window (src/MG_solver_CPU.cpp:156..429).  BASELINE.md section 3: both builds (-O2 and the shipped
    Makefile's flags, src/Makefile:8: no -O), all of the job's cores and one thread, >= 3 repetitions with
    min/median for the headline variant; bounded to ~20 s of CPU work.
    TEST/BASELINE infrastructure: reported beside the GPU number, never part of it."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle
    orc = _oracle.Oracle()
    cores = min(16, os.cpu_count() or 1)  # a 1-GPU job owns 16 host cores on the GPU box, whatever os.cpu_count() says
    have_ref = _oracle.have_reference()
    kind = "reference" if have_ref else "port"

    def timed(n, threads, reps, makefile_flags=False):
        path = os.path.join(tmp, f"cpu_{cycle}cycle_{n}.txt")
        write_vcycle_file(path, n, n_min, nu, 1e-7)
        szs = level_sizes(n, n_min)
        lups = sum(2 * nu * v * s * s for s, v in zip(szs[:-1], level_visits(szs, cycle)[:-1]))
        ops = None
        if have_ref:
            ops = _oracle.Reference(makefile_flags=makefile_flags)
            ops.set_threads(threads)
        orc.set_threads(threads)
        ts, err = [], None
        for _ in range(reps):
            res = orc.run_cycle_file(path, ops=ops, want_report=False)
            if res["status"] != 0:
                return None
            ts.append(res["time_ms"])
            err = res["mg_error"]
        return {"N": n, "threads": threads, "reps": reps, "min_ms": round(min(ts), 2), "median_ms": round(statistics.median(ts), 2),
                "value": round(lups / (min(ts) * 1e-3) / 1e6, 3), "unit": "MLUPS", "mg_error": err}

    reps = 3 if cycle == "V" else 1   # (a W-cycle is 1.5x the sweeps of a V-cycle plus 2^k coarse solves: one run bounds the leg)
    head = timed(N, cores, reps)
    if not head:
        return None
    out = {"value": head["value"], "unit": "MLUPS", "cores": cores, "kind": kind, "time_ms": head["min_ms"],
           "median_ms": head["median_ms"], "reps": reps, "mg_error": head["mg_error"],
           "sample": (f"{cycle}({nu},{nu})-cycle at N={N}^2, the reference's timed window, {reps} run{'s (value = fastest)' if reps > 1 else ''}, "
                      f"{'reference operators (oracle/_ref/libmgref.so, g++ -O2 -fopenmp)' if have_ref else 'oracle restatement (-O2)'}"
                      f", {cores} OpenMP threads"),
           "variants": {}}
    small = min(N, 4096)
    if have_ref and os.path.exists(_oracle.REF_SO_MAKEFLAGS):
        v = timed(N, cores, 1, makefile_flags=True)
        if v:
            v["build"] = "g++ -fopenmp (src/Makefile:8: no -O)"
            out["variants"]["makefile_flags"] = v
    v = timed(small, 1, 1)
    if v:
        v["build"] = "g++ -O2 -fopenmp" if have_ref else "oracle -O2"
        out["variants"]["one_thread"] = v
    return out


def self_launch(args):
    """--gpus N > 1 outside torchrun: start the N ranks as a CHILD process (never exec: this interpreter may be
    preloaded by a profiler) before anything in this process has touched the GPU, relay rank 0's JSON line."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MG_BENCH_N"] = str(args.n)  # torchrun's own parser would eat a bare --n
    child = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in child.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line:
        print(line, flush=True)
    return child.returncode if child.returncode != 0 else (0 if line else 1)


def time_cycle(mg, plan, steps, warmup, profile_min_N=None, first=None):
    """W untimed runs, then exactly K windows enqueued back to back on the engine's stream (a fixed-step cycle
    file needs no per-step host sync), bracketed by synchronisation; optional live hipEvent pairs.
    first: the result of the plan's first run when the caller made it earlier (pool, tables, code objects)."""
    if first is None:
        first = plan.execute()   # (pool, tables, code objects)
    assert first["status"] == 0, first
    time_cycle.first_ms = first["device_ms"]
    # the W untimed steps run exactly like the timed ones -- enqueued back to back -- so the clocks are where a
    # sustained run holds them when the timed region starts (measured: the first 20 windows after an idle gap take
    # 0.80-0.82 ms, from the ~40th on 0.735)
    for _ in range(warmup):
        plan.enqueue()
    mg.sync()
    r0 = plan.collect()
    assert r0["status"] == 0, r0
    if profile_min_N is not None:
        # live hipEvent pairs around the finest-level launches of every 5th timed window (a pair costs its launch
        # ~2.5 us: 10 us on a 0.74 ms window with both fine-level launches bracketed in every window)
        mg.profile_begin(min_N=profile_min_N, every=5 if steps >= 10 else 1)
    t0 = time.perf_counter()
    for _ in range(steps):
        plan.enqueue()
    mg.sync()
    t1 = time.perf_counter()
    r = plan.collect()
    prof = mg.profile_end() if profile_min_N is not None else []
    assert r["status"] == 0, r
    return (t1 - t0) * 1e3 / steps, r, prof


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--n", type=int, default=int(os.environ.get("MG_BENCH_N", "8192")),
                    help="finest grid size N (N x N points); under torchrun use env MG_BENCH_N (its parser eats --n)")
    ap.add_argument("--n-min", type=int, default=8)
    ap.add_argument("--nu", type=int, default=3, help="smoothing steps per node (V(nu,nu))")
    ap.add_argument("--cycle", choices=["V", "W"], default="V")
    ap.add_argument("--mode", choices=["eager", "graph", "unfused"], default="eager")
    ap.add_argument("--smoother", choices=["stream", "simple"], default=os.environ.get("MG_SMOOTHER", "stream"))
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-strong", action="store_true", help="skip the N=16384 strong-scaling leg")
    ap.add_argument("--no-large", action="store_true", help="skip the N=32768 leg (a second strong-scaling base, nested as strong_scaling_32768)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=os.environ.get("MG_BENCH_SCALING", "weak"),
                    help="which leg is the line's `value` (the other one is nested)")
    ap.add_argument("--mixed", action="store_true",
                    help="mixed-precision mode (fp32 cycle, fp64 source/result; NOT the headline metric, which is fp64)")
    ap.add_argument("--refine", type=int, default=1,
                    help="with --mixed: fp32 cycles per window, joined by the fp64 residual of the fp64 iterate and an fp64 correction "
                         "(BASELINE.json configs[4]: fp32 smoothing / fp64 residual correction)")
    ap.add_argument("--force-slab", action="store_true",
                    help="run the row-slab/RCCL leg even with one rank (plumbing rehearsal on a 1-GPU box)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))  # nothing above has touched the GPU

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    from multigrid_poisson_solver_amd import build as mg_build
    mg_build.ensure_built()   # no-op when the in-tree library is there (the normal case)
    if world > 1 or args.force_slab:
        import bench_multi  # row-slab path (one process per GPU, RCCL ghost rows)
        return bench_multi.run(args, rank, world, local_rank)

    import multigrid_poisson_solver_amd as mg
    mg.init(local_rank)
    mg.set_smoother(args.smoother)

    N, nu = args.n, args.nu
    sizes = level_sizes(N, args.n_min)
    tmp = tempfile.mkdtemp(prefix="mgbench_")
    cyc = os.path.join(tmp, f"{args.cycle}cycle_{N}.txt")
    write_cycle = mg.write_vcycle_file if args.cycle == "V" else mg.write_wcycle_file
    write_cycle(cyc, N, args.n_min, nu, 1e-7)
    visits = level_visits(sizes, args.cycle)
    # lattice updates of one step: (nu1+nu2) * sum over smoothed levels of n_l (x visits for W)
    lups = sum(2 * nu * v * s * s for s, v in zip(sizes[:-1], visits[:-1]))
    # the W-cycle's algorithmic bytes are the per-level term weighted by the level's visits
    algo_bytes = sum(v * ((8 + 24 * nu + 24 + 8 + 16 + 24 * nu) * a * a + 16.0 * b * b) for v, a, b in zip(visits, sizes[:-1], sizes[1:]))

    # the headline plan is made and run ONCE before anything else (its arrays, tables and code objects exist from here on):
    # between the sustained load of the legs below and its own windows nothing is left to do on the host
    refine = max(1, args.refine) if args.mixed else 1
    plan = mg.CyclePlan(cyc, fused=(args.mode != "unfused"), graph=(args.mode == "graph"), report=False, error=False,
                        mixed=args.mixed, refinement=refine)
    first_run = plan.execute()
    held = []   # the legs' plans stay open until the headline is timed (closing one frees GiBs: an idle gap for the device)
    # The strong-scaling bases run FIRST: besides being legs of the line they are ~0.3 s of sustained load, after which
    # the headline leg -- W untimed and K timed windows exactly as passed -- starts at the clocks a sustained run holds
    # (after an idle gap the device needs ~40 windows to get there: profiles/r03_warmup.txt; `clock_state` in the line
    # says what preceded the timed region and how the first and the last window of this leg compare).
    pre_legs, windows_before = {}, 0
    # ---- strong-scaling base: the V-cycle of BASELINE.json configs[3] (N = 16384^2) on this one GPU ----
    if not args.no_strong and args.cycle == "V" and N != STRONG_N:
        try:
            scyc = os.path.join(tmp, f"Vcycle_{STRONG_N}.txt")
            mg.write_vcycle_file(scyc, STRONG_N, args.n_min, nu, 1e-7)
            ssizes = level_sizes(STRONG_N, args.n_min)
            slups = sum(2 * nu * s * s for s in ssizes[:-1])
            splan = mg.CyclePlan(scyc, fused=True, report=False, error=False, mixed=args.mixed)
            sms, sr, _ = time_cycle(mg, splan, args.steps, max(1, args.warmup))
            windows_before += 1 + max(1, args.warmup) + args.steps
            scb = vcycle_compulsory_bytes(ssizes) * (0.5 if args.mixed else 1.0)
            pre_legs["strong_scaling"] = {"N": STRONG_N, "n_gpus": 1, "ms_per_step": round(sms, 4), "value": round(slups / (sms * 1e-3) / 1e6, 1),
                                     "unit": "MLUPS", "steps": args.steps, "scaling": "strong",
                                     "workload": f"V({nu},{nu})-cycle N={STRONG_N}^2 fp64 (BASELINE.json configs[3]), one GPU: the base of the strong-scaling curve",
                                     "cycle_frac_of_hbm_peak": round(scb / (sms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            held.append(splan)
        except mg.MGError as exc:
            pre_legs["strong_scaling"] = {"error": str(exc)}

    # ---- the same one level up: N = 32768^2 fp64 (few windows: 10 ms each) ----
    if not args.no_strong and not args.no_large and args.cycle == "V" and not args.mixed and N != LARGE_N:
        try:
            lcyc = os.path.join(tmp, f"Vcycle_{LARGE_N}.txt")
            mg.write_vcycle_file(lcyc, LARGE_N, args.n_min, nu, 1e-7)
            lsizes = level_sizes(LARGE_N, args.n_min)
            llups = sum(2 * nu * s * s for s in lsizes[:-1])
            lplan = mg.CyclePlan(lcyc, fused=True, report=False, error=False)
            lsteps = min(args.steps, 10)
            lms, _, _ = time_cycle(mg, lplan, lsteps, min(max(1, args.warmup), 10))
            windows_before += 1 + min(max(1, args.warmup), 10) + lsteps
            pre_legs["strong_scaling_32768"] = {"N": LARGE_N, "n_gpus": 1, "ms_per_step": round(lms, 4), "value": round(llups / (lms * 1e-3) / 1e6, 1),
                                           "unit": "MLUPS", "steps": lsteps, "scaling": "strong",
                                           "workload": f"V({nu},{nu})-cycle N={LARGE_N}^2 fp64, one GPU: base of a second strong-scaling curve",
                                           "cycle_frac_of_hbm_peak": round(vcycle_compulsory_bytes(lsizes) / (lms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            held.append(lplan)
        except mg.MGError as exc:
            pre_legs["strong_scaling_32768"] = {"error": str(exc)}

    # ---- the headline cycle file replayed from a hipGraph (MG_CYCLE_GRAPH): a leg of its own -- is the captured window
    # faster than the eager one? -- and, run last before the headline, 50 windows of the very workload that is timed next ----
    if not args.no_strong and args.mode == "eager" and not args.mixed:
        try:
            gplan = mg.CyclePlan(cyc, fused=True, graph=True, report=False, error=False)
            gsteps = 50
            gms, gr, _ = time_cycle(mg, gplan, gsteps, 3)
            windows_before += 1 + 3 + gsteps
            pre_legs["graph_replay"] = {"N": N, "ms_per_step": round(gms, 4), "value": round(lups / (gms * 1e-3) / 1e6, 1), "unit": "MLUPS",
                                        "steps": gsteps, "graph_replayed": bool(gr.get("graph_replayed")),
                                        "workload": f"the headline {args.cycle}({nu},{nu})-cycle at N={N}^2 captured into a hipGraph and replayed (MG_CYCLE_GRAPH)"}
            held.append(gplan)
        except mg.MGError as exc:
            pre_legs["graph_replay"] = {"error": str(exc)}

    lups *= refine   # a window of the refinement runs the cycle file `refine` times (fp32), joined by an fp64 residual + correction
    # ---- timed region: exactly K steps, bracketed by synchronisation on both sides ----
    ms_per_step, r, prof = time_cycle(mg, plan, args.steps, max(args.warmup, 2 if args.mode == "graph" else 0), profile_min_N=N, first=first_run)
    for h in held:
        h.close()
    dev_ms = r["device_ms"]
    mlups = lups / (ms_per_step * 1e-3) / 1e6
    elem = 0.5 if args.mixed else 1.0

    # dominant kernel = the launch family with the largest total time on the finest grid
    roof = None
    kernels = []
    for e in sorted(prof, key=lambda e: -e["total_ms"]):
        avg = e["total_ms"] / max(1, e["launches"])
        cb = compulsory_bytes(e["name"], N)
        kernels.append({"kernel": e["name"], "N": e["N"], "launches": e["launches"], "avg_ms": round(avg, 4),
                        "compulsory_GBs": round(cb * elem / (avg * 1e-3) / 1e9, 1) if cb and avg > 0 else None,
                        "algorithmic_equiv_GBs": round(e["algo_bytes"] / (avg * 1e-3) / 1e9, 1) if avg > 0 else None})
    if kernels and kernels[0]["compulsory_GBs"]:
        k0 = kernels[0]
        cb = compulsory_bytes(k0["kernel"], N) * elem
        # `achieved`/`frac`: the bytes the launch must move (every input read once, every output written once)
        # over its measured duration -- a physical bandwidth.  SURVEY section 8d's per-sweep accounting
        # (24 B per lattice update and sweep) is kept under `algorithmic_equiv`: the temporally blocked kernel
        # does S sweeps per pass over HBM, so that figure exceeds the peak and is a saving, not a bandwidth.
        roof = {"bound": "hbm", "kernel": k0["kernel"], "achieved": k0["compulsory_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(k0["compulsory_GBs"] / HBM_PEAK_GBS, 4), "compulsory_bytes": cb,
                "traffic": measured_traffic(k0["kernel"]) if not args.mixed else None,
                "avg_ms": k0["avg_ms"], "launches": k0["launches"],
                **({"note": "fused `1` node that RECOMPUTES the level's pre-smoothed field (3 sweeps from zero) instead of reading it: "
                            "8 B/pt less to read here, 8 B/pt less to write in the `-1` node, twice the arithmetic -- the launch is "
                            "co-limited by VALU issue, so its byte rate is below the 0.60-0.62 of the store/re-read form while "
                            "the cycle is 15 % faster (DESIGN 3.1)",
                    # the same launch priced at the bytes of the form it replaces (U read as well: 24 n + 8 m)
                    "store_reread_equiv": {"bytes": (24.0 * N * N + 8.0 * (N // 2) ** 2) * elem,
                                           "GBs": round((24.0 * N * N + 8.0 * (N // 2) ** 2) * elem / (k0["avg_ms"] * 1e-3) / 1e9, 1),
                                           "frac": round((24.0 * N * N + 8.0 * (N // 2) ** 2) * elem / (k0["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
                   if "pre3" in k0["kernel"] else {}),
                "algorithmic_equiv": {"bytes": next(e["algo_bytes"] for e in prof if e["name"] == k0["kernel"]),
                                      "GBs": k0["algorithmic_equiv_GBs"],
                                      "note": "SURVEY 8d bytes (one HBM pass per sweep) / launch time; > peak because S sweeps share one pass"}}

    out = {
        "metric": "vcycle_mlups", "value": round(mlups, 1), "unit": "MLUPS", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32" if args.mixed else "f64", "data": "synthetic",
        "config": {"workload": f"{args.cycle}({nu},{nu})-cycle N={N}^2 {'fp32 cycle (mixed mode)' if args.mixed else 'fp64'}, {len(sizes)} levels to N={sizes[-1]}, "
                               f"red-black GS(1e-7) coarse solve, cycle-file driver ({args.mode}, {args.smoother} smoother: "
                               f"temporally blocked wave-streaming kernels on the large levels, register-tile kernels on the levels 65..1024, "
                               f"one-workgroup LDS kernel below; halos recomputed, neighbours via DPP lane shifts)"
                               + (f", {refine} fp32 cycles per window joined by fp64 residual + correction" if refine > 1 else ""),
                   "N": N, "levels": len(sizes), "cycle_file": os.path.basename(cyc)},
        "device_ms_per_step": round(dev_ms, 4),
        "fine_dof_per_s": round(N * N / (ms_per_step * 1e-3), 1),
        "mg_error": plan.analytic_error(r),
        "roofline": roof,
        "kernels": kernels[:8],
        # what the device had been doing when the timed region started (windows of all legs, this one's first run and its
        # W untimed ones) and the device time of this leg's first and last window
        "clock_state": {"windows_before_timed": windows_before + 1 + max(args.warmup, 2 if args.mode == "graph" else 0),
                        "first_window_ms": round(time_cycle.first_ms, 4), "last_window_ms": round(dev_ms, 4)},
    }
    out.update(pre_legs)
    if algo_bytes:
        # (a W-cycle runs as a batched schedule, whose levels use the recomputing pair from N = 1024 on)
        cb = vcycle_compulsory_bytes(sizes, visits=visits, min_n=BATCH_RECOMPUTE_MIN_N if args.cycle == "W" and not args.mixed else None) * elem * refine
        if refine > 1:   # per joint: fp64 iterate + fp64 source read, fp32 source written; fp32 correction read, fp64 iterate read + written
            cb += (refine - 1) * (8 + 8 + 4 + 4 + 8 + 8) * float(N) * N
        algo_bytes *= refine
        out["cycle_roofline"] = {"compulsory_bytes": cb, "achieved": round(cb / (ms_per_step * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": round(cb / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                 "algorithmic_equiv": {"bytes": algo_bytes, "GBs": round(algo_bytes / (ms_per_step * 1e-3) / 1e9, 1)}}
    plan.close()
    mg.lib().mg_pool_trim()

    # north_star's own yardstick, measured beside the cycle: ONE fine-level Jacobi sweep per launch (doSmoothing with
    # step = 1, 24 B per point: compulsory == algorithmic here) -- the one-row-per-block pair kernel that the engine
    # routes a bare sweep of a large grid to, and the streaming kernel at S = 1 (smoother "stream_only")
    try:
        Ua, Ub, Ff = mg.DeviceGrid.uniform(N, 3), mg.DeviceGrid(N), mg.DeviceGrid.uniform(N, 7)
        single = {}
        for sm in ("stream", "stream_only"):
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
            single["k_" + e["name"].split("<")[0] + ("_S1" if "stream" in e["name"] else "")] = {
                "launch": e["name"], "avg_ms": round(avg, 4), "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
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

    if args.scaling == "strong" and isinstance(out.get("strong_scaling"), dict) and "value" in out["strong_scaling"]:
        s = out["strong_scaling"]
        out["weak_scaling"] = {"N": N, "value": out["value"], "ms_per_step": out["ms_per_step"]}
        out.update(value=s["value"], ms_per_step=s["ms_per_step"], scaling="strong")
        out["config"]["workload"] = s["workload"]
        out["config"]["N"] = STRONG_N

    if not args.no_cpu:
        base = cpu_baseline(tmp, N, args.n_min, nu, write_cycle, args.cycle)
        if base:
            out["cpu_baseline"] = base
    mg.finalize()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
