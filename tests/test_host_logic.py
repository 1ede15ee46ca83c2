"""CPU-only checks of the product's host side: the C-ABI library loads and exports every
symbol include/mg_hip.h declares, the host-built R/P tables equal the oracle's and the
committed goldens, the engine refuses to run without a GPU (no CPU fallback), and the
cycle-file generators reproduce the shipped files.  No compute call is made here."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, assert_bits


@pytest.fixture(scope="module")
def m():
    import multigrid_poisson_solver_amd as m
    if not os.path.exists(m.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    m.load_library()
    return m


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mg_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mg_[A-Za-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(m):
    import ctypes
    lib = ctypes.CDLL(m.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), f"libmgpoisson.so does not export {n}"
    # and the Python binding table covers the same set
    assert sorted(m.ABI) == names


def test_dropin_header_names_match_reference_surface():
    """include/mg_dropin.hpp maps the reference's six operator names (plus the problem
    functions) onto the ABI with identical argument lists (src/MG_solver_CPU.cpp:16-30)."""
    text = open(os.path.join(ROOT, "include", "mg_dropin.hpp")).read()
    for proto in ["void getResidual(int N, double L, double* U, double* F, double* D)",
                  "void doGridAddition(int N, double* U1, double* U2)",
                  "void doSmoothing(int N, double L, double* U, double* F, int step, double* error)",
                  "void doExactSolver(int N, double L, double* U, double* F, double target_error, int option)",
                  "void doRestriction(int N, double* U_f, int M, double* U_c)",
                  "void doProlongation(int N, double* U_c, int M, double* U_f)"]:
        assert proto in text, proto


def test_no_cpu_fallback_without_gpu(m):
    """On a machine without a HIP device mg_init must fail loudly and every operator must
    refuse to run (the product path never routes through a CPU implementation)."""
    # (not torch.cuda.is_available(): importing torch AFTER libmgpoisson.so would load a second
    # HIP runtime next to the one the engine is linked against)
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present")
    code = ("import multigrid_poisson_solver_amd as m\n"
            "try:\n    m.init(0)\nexcept m.MGError as e:\n    print('REFUSED', e)\n"
            "lib = m.load_library(); lib.mg_set_abort_on_error(0)\n"
            "lib.mg_doSmoothing(16, 1.0, None, None, 1, None)\nprint('ERR', lib.mg_last_error())\n")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert "REFUSED" in out.stdout and "no CPU fallback" in (out.stdout + out.stderr)
    assert "ERR 4" in out.stdout


def test_cli_argument_errors(m):
    """src/MG_solver_CPU.cpp:51-54: wrong argument count prints the reference's message
    and exits 1."""
    out = subprocess.run([m.EXE_PATH], capture_output=True, text=True)
    assert out.returncode == 1 and "[ ERROR ]: Wrong input numbers of parameter." in out.stdout


def test_product_restriction_tables(m, oracle, golden_tables):
    N = 32768
    while N // 2 >= 8:
        M = N // 2
        lo, w = m.restriction_table(N, M)
        assert np.array_equal(lo, golden_tables[f"rt_lo_{N}to{M}"])
        assert_bits(w, golden_tables[f"rt_w_{N}to{M}"], f"restriction weights {N}->{M}")
        N = M
    for (N, M) in [(17, 9), (16, 15), (33, 16), (100, 37), (256, 255), (11585, 5792)]:
        lo, w = m.restriction_table(N, M)
        olo, ow = oracle.restriction_table(N, M)
        assert np.array_equal(lo, olo)
        assert_bits(w, ow, "w")


def test_product_prolongation_tables(m, oracle, golden_tables):
    N = 32768
    while N // 2 >= 8:
        M = N // 2
        for axis in (0, 1):
            owner, hi, lo = m.prolongation_table(M, N, axis)
            assert np.array_equal(owner, golden_tables[f"po_{M}to{N}"]), (M, N, axis)
        N = M
    for (N, M) in [(15, 16), (9, 17), (16, 33), (4, 8), (37, 100), (5792, 11585)]:
        want = oracle.prolongation_owner(N, M)
        for axis in (0, 1):
            owner, hi, lo = m.prolongation_table(N, M, axis)
            assert np.array_equal(owner, want)
            assert owner.min() >= 0 and owner.max() <= N - 2
            # the two 1-D weights of one fine index sum to the coarse spacing (to rounding)
            np.testing.assert_allclose(hi + lo, 1.0 / (N - 1), rtol=1e-12)


def test_product_prolongation_table_reproduces_oracle_values(m, oracle):
    """Evaluate the gather formula of the prolongation kernel on the host with the product's
    tables: it must equal the oracle's (= the reference's) scatter loop bit for bit."""
    for (N, M) in [(4, 8), (8, 16), (15, 16), (9, 17), (16, 33), (32, 64)]:
        rng = np.random.default_rng(N * 1000 + M)
        Uc = rng.random((N, N)) - 0.5
        orow, rhi, rlo = m.prolongation_table(N, M, 0)
        ocol, chi, clo = m.prolongation_table(N, M, 1)
        c_dx = 1.0 / float(N - 1)
        I, J = np.meshgrid(orow, ocol, indexing="ij")
        c1, c2, c3, c4 = Uc[I, J], Uc[I, J + 1], Uc[I + 1, J], Uc[I + 1, J + 1]
        xh, xl = chi[None, :], clo[None, :]
        yh, yl = rhi[:, None], rlo[:, None]
        mine = ((c1 * xh + c2 * xl) * yh + (c3 * xh + c4 * xl) * yl) / c_dx / c_dx
        assert_bits(mine, oracle.doProlongation(N, Uc, M, fill=0.0), f"prolongation tables {N}->{M}")


def test_cycle_file_generators(m, tmp_path):
    v = tmp_path / "v.txt"
    m.write_vcycle_file(str(v), 256, 8, 3, 1e-7)
    import _cycles
    shipped = _cycles.text("Vcycle.txt").split()
    assert [float(t) for t in v.read_text().split()] == [float(t) for t in shipped]
    w = tmp_path / "w.txt"
    m.write_wcycle_file(str(w), 256, 8, 3, 1e-8, depth=4)
    shipped = _cycles.text("Wcycle.txt").split()
    assert [float(t) for t in w.read_text().split()] == [float(t) for t in shipped]
    assert m.write_vcycle_file(str(v), 8192, 8) == 11


def test_weak_scaling_grid_sizes():
    """bench.py --gpus N: ~8192^2 points per GPU on grids m * 2^j (m <= 64), so that every level above the
    coarse-tail kernel (N <= 64) has an even size and every node runs in its fused one-launch form."""
    import bench
    assert [bench.grid_for(w) for w in (1, 2, 4, 8)] == [8192, 11520, 16384, 23040]
    for w in range(1, 9):
        N = bench.grid_for(w)
        assert abs(N * N / w - 8192 * 8192) < 0.03 * 8192 * 8192
        sizes = bench.level_sizes(N, 8)
        assert all(n % 2 == 0 for n in sizes if n > 64), sizes
        assert sizes[-1] >= 8 and sizes[-1] ** 2 <= 256   # the one-wave coarse solvers cover it


def test_recompute_pair_and_closed_form_owner_predicates(m):
    """Host-side predicates the kernels rely on (no device needed): the recomputing node pair exists for every pair of at most
    six sweeps together (1..4 each) in the default build (mg_recompute_pair_available); and the closed form the register-tile kernel uses for the
    owner cell of the fused prolongation, min(k*(Nc-1)/(N-1), Nc-2), reproduces the tables built from the reference's
    ceil() expressions for the level pairs of halving hierarchies (ProlongTable::closed_form verifies this entry by
    entry at run time; here the same comparison on the host for a spread of sizes, incl. an odd fine size where the
    kernel would read the tables)."""
    lib = m.load_library()
    assert [lib.mg_recompute_pair_available(s, s) for s in (1, 2, 3, 4)] == [1, 1, 1, 0]
    assert [[lib.mg_recompute_pair_available(a, b) for b in (1, 2, 3, 4)] for a in (1, 2, 3, 4)] == \
        [[1, 1, 1, 1], [1, 1, 1, 1], [1, 1, 1, 0], [1, 1, 0, 0]]   # pre + post <= 6
    assert lib.mg_recompute_pair_available(0, 3) == 0 and lib.mg_recompute_pair_available(5, 1) == 0
    for Nc, N in [(64, 128), (128, 256), (512, 1024), (1024, 2048), (362, 724), (45, 90), (352, 704), (90, 181), (33, 64), (2048, 4096)]:
        k = np.arange(N, dtype=np.int64)
        want = np.minimum(k * (Nc - 1) // (N - 1), Nc - 2)
        for axis in (0, 1):
            o = np.empty(N, dtype=np.int32)
            a, b = np.empty(N), np.empty(N)
            lib.mg_prolongation_table(Nc, N, axis, o.ctypes.data, a.ctypes.data, b.ctypes.data)
            assert np.array_equal(o, want), (Nc, N, axis)


def test_bench_accounting_helpers():
    """bench.py's byte and update counts: a V-cycle visits every level once, the W-cycle of the shipped
    recursion visits level l >= 1 2^l times; the W-cycle's compulsory bytes are the per-level terms weighted by the visits."""
    sys.path.insert(0, ROOT)
    import bench
    sizes = bench.level_sizes(8192, 8)
    assert sizes == [8192 >> l for l in range(11)]
    assert bench.level_visits(sizes, "V") == [1] * 11
    assert bench.level_visits(sizes, "W") == [1] + [2 ** l for l in range(1, 11)]
    v = bench.vcycle_compulsory_bytes(sizes)
    w = bench.vcycle_compulsory_bytes(sizes, visits=bench.level_visits(sizes, "W"))
    # recomputing pair (levels >= MG_RECOMPUTE_MIN_N: 4096 in the product, lower in this suite): 24 n + 16 m; 40 n + 16 m below; nothing for the LDS tail
    assert v == pytest.approx(sum((24.0 if a >= bench.RECOMPUTE_MIN_N else 40.0) * a * a + 16.0 * (a // 2) ** 2 for a in sizes[:-1] if a > 64))
    assert v < w < 2.0 * v   # (the geometric sum of 2^l / 4^l)
    assert bench.vcycle_algorithmic_bytes(sizes, 3, 3) == pytest.approx(sum(200.0 * a * a + 16.0 * (a // 2) ** 2 for a in sizes[:-1]))


def test_committed_bench_line_keeps_the_contract():
    """profiles/rNN_bench_line.json (the newest round's) is what bench.py printed on the MI355X: the keys the driver and the
    judge read must be there; `roofline.frac` is PHYSICAL (compulsory bytes / time / peak, below 1), the per-sweep accounting
    of SURVEY 8d sits beside it as `algorithmic_equiv`; the counter-measured traffic is tied to the profiled library's sha."""
    import glob
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    newest = sorted(glob.glob(os.path.join(root, "profiles", "r[0-9][0-9]_bench_line.json")))[-1]
    d = json.load(open(newest))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "strong_scaling", "single_sweep_roofline"):
        assert k in d, k
    assert d["unit"] == "MLUPS" and d["dtype"] == "f64" and d["data"] == "synthetic" and d["n_gpus"] == 1
    assert "workload" in d["config"] and "8192" in d["config"]["workload"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.4 < r["frac"] < 0.8
    assert abs(r["achieved"] - r["compulsory_bytes"] / (r["avg_ms"] * 1e-3) / 1e9) < 1.0
    assert r["algorithmic_equiv"]["GBs"] > r["peak"]                    # S sweeps per pass: a saving, not a bandwidth
    t = r["traffic"]
    assert t is not None and len(t["lib_sha"]) == 16 and 0.95 < t["bytes"] / r["compulsory_bytes"] < 1.15   # counters vs compulsory bytes
    assert set(d["single_sweep_roofline"]) == {"k_jacobi_pair", "k_jacobi_stream_S1"}
    lups = 6 * sum((8192 >> l) ** 2 for l in range(10))
    assert abs(d["value"] - lups / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-3 * d["value"]
    s = d["strong_scaling"]
    assert s["N"] == 16384 and s["n_gpus"] == 1 and s["value"] > 0
    if "strong_scaling_32768" in d:   # (from round 3 on: the second strong-scaling base)
        assert d["strong_scaling_32768"]["N"] == 32768 and d["strong_scaling_32768"]["value"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and "sample" in c and c["value"] > 0 and c["reps"] == 3
    assert {"makefile_flags", "one_thread"} <= set(c["variants"])


def test_exp_table_and_algorithm_reproduce_this_libm():
    """getSource runs on the device with libm's exp() algorithm (mg_kernels.hip:exp_libm; table =
    csrc/mg_exp_table.h, generated by scripts/gen_exp_table.py from its definition).  CPU check of both: the
    committed header is what the generator prints, and the algorithm -- restated here with exact rational arithmetic
    for every fused multiply-add -- returns math.exp's bits on arguments of the kind getSource produces (x - y on
    the unit square and around it).  On a host whose libm is another one this test fails and the engine's run-time
    self-check keeps the host form of getSource."""
    import math
    import re
    import struct
    import subprocess
    import sys
    from fractions import Fraction
    gen = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gen_exp_table.py")], capture_output=True, text=True, check=True).stdout
    hdr = open(os.path.join(ROOT, "multigrid_poisson_solver_amd", "csrc", "mg_exp_table.h")).read()
    assert gen == hdr
    tab = [int(v, 16) for v in re.findall(r"0x([0-9a-f]{16})ull", hdr)]
    assert len(tab) == 256

    def f2b(x):
        return struct.unpack("<Q", struct.pack("<d", x))[0]

    def b2f(b):
        return struct.unpack("<d", struct.pack("<Q", b & (2 ** 64 - 1)))[0]

    def fma(a, b, c):  # one rounding: float(Fraction) rounds to nearest even
        return float(Fraction(a) * Fraction(b) + Fraction(c))

    InvLn2N, Shift = float.fromhex("0x1.71547652b82fep+7"), float.fromhex("0x1.8p+52")
    NegLn2hiN, NegLn2loN = float.fromhex("-0x1.62e42fefa0000p-8"), float.fromhex("-0x1.cf79abc9e3b3ap-47")
    C2, C3 = float.fromhex("0x1.ffffffffffdbdp-2"), float.fromhex("0x1.555555555543cp-3")
    C4, C5 = float.fromhex("0x1.55555cf172b91p-5"), float.fromhex("0x1.1111167a4d017p-7")

    def exp_alg(x):
        abstop = (f2b(x) >> 52) & 0x7ff
        if abstop < 0x3c9:
            return 1.0 + x
        assert abstop - 0x3c9 <= 0x3e
        kd_s = fma(x, InvLn2N, Shift)
        ki = f2b(kd_s)
        kd = kd_s - Shift
        r = fma(kd, NegLn2hiN, x)
        r = fma(kd, NegLn2loN, r)
        idx = 2 * (ki & 127)
        tail, sbits = b2f(tab[idx]), tab[idx + 1] + (ki << 45)
        p23, rt, r2, p45 = fma(r, C3, C2), r + tail, r * r, fma(r, C5, C4)
        t = fma(p23, r2, rt)
        tmp = fma(r2 * r2, p45, t)
        scale = b2f(sbits)
        return fma(scale, tmp, scale)

    rng = np.random.default_rng(7)
    args = list(rng.uniform(-1.0, 1.0, 1500)) + list(rng.uniform(-4.0, 4.0, 300)) + [0.0, -0.0, 1e-300, 0.5, -0.5, 1.0 / 8191, -2047.0 / 8191]
    N = 129
    h = 1.0 / (N - 1)
    args += [c * h - r * h for r in (1, 7, 64, 127) for c in range(1, N - 1, 5)]
    bad = [x for x in args if f2b(exp_alg(x)) != f2b(math.exp(x))]
    assert not bad, f"{len(bad)} of {len(args)} arguments differ from math.exp, e.g. {bad[:3]}"
