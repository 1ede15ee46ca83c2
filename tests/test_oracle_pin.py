"""The oracle is pinned twice: against the committed golden vectors (always) and against
the reference's own code compiled from /root/reference (when oracle/_ref is built)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import _oracle
import _synth
from conftest import GOLDEN, assert_bits


def lcg_field(N, seed):
    return _oracle.lcg_uniform(N * N, seed).reshape(N, N)


@pytest.mark.parametrize("N", [16, 64])
def test_oracle_vs_golden_ops(oracle, golden_ops, N):
    g = golden_ops
    assert_bits(oracle.getSource(N), g[f"N{N}_F_source"], "getSource")
    assert_bits(oracle.getAnalytic(N), g[f"N{N}_analytic"], "getAnalytic")
    # the generator's seeded inputs are reproducible
    assert_bits(lcg_field(N, 0x9E3779B97F4A7C15 + N), g[f"N{N}_U_rand"], "lcg U")
    for tag, U0, F in (("src", np.zeros((N, N)), g[f"N{N}_F_source"]), ("rnd", g[f"N{N}_U_rand"], g[f"N{N}_F_rand"])):
        for s in (1, 3, 10):
            U, e = oracle.doSmoothing(N, 1.0, U0, F, s)
            assert_bits(U, g[f"N{N}_{tag}_smooth{s}_U"], f"smoothing {tag} {s}")
            assert e == pytest.approx(g[f"N{N}_{tag}_smooth{s}_err"][0], rel=1e-12)
        U3 = g[f"N{N}_{tag}_smooth3_U"]
        D = oracle.getResidual(N, 1.0, U3, F)
        assert_bits(D, g[f"N{N}_{tag}_residual"], "residual")
        assert_bits(oracle.doRestriction(N, -D, N // 2), g[f"N{N}_{tag}_restrict_negD"], "restrict(-D)")
        assert_bits(oracle.doProlongation(N // 2, U3[: N // 2, : N // 2].copy(), N), g[f"N{N}_{tag}_prolong_from_half"], "prolong")
        assert_bits(oracle.doGridAddition(N, U3, F), g[f"N{N}_{tag}_add"], "add")


def test_oracle_vs_golden_odd_pairs(oracle, golden_ops):
    g = golden_ops
    for (Nf, Mc) in ((17, 9), (16, 15), (33, 16)):
        assert_bits(oracle.doRestriction(Nf, g[f"restrict_{Nf}to{Mc}_in"], Mc), g[f"restrict_{Nf}to{Mc}_out"], f"restrict {Nf}->{Mc}")
    for (Nc, Mf) in ((15, 16), (9, 17), (16, 33)):
        assert_bits(oracle.doProlongation(Nc, g[f"prolong_{Nc}to{Mf}_in"], Mf, fill=0.0), g[f"prolong_{Nc}to{Mf}_out"], f"prolong {Nc}->{Mf}")


@pytest.mark.parametrize("N", [8, 16, 17])
def test_oracle_vs_golden_exact_solver(oracle, golden_ops, N):
    assert_bits(oracle.doExactSolver(N, 1.0, golden_ops[f"gs_N{N}_F"], 1e-7), golden_ops[f"gs_N{N}_U"], "GaussSeidel")


def test_reference_kats(oracle, golden_ops):
    """Analytic known-answer tests of the reference's own test programs: bilinear
    sampling of a linear ramp is exact.  testFunction/Test_doRestriction_GPU.cu:189-193,
    testFunction/Test_doProlongation_GPU.cu:190-194."""
    ramp16 = np.add.outer(np.arange(16.0), np.arange(16.0))
    out = oracle.doRestriction(16, ramp16, 8)
    assert_bits(out, golden_ops["kat_restrict_16to8"], "KAT restrict golden")
    ix = np.arange(8.0)
    expect = np.add.outer(ix, ix) * 15.0 / 7.0
    np.testing.assert_allclose(out[1:-1, 1:-1], expect[1:-1, 1:-1], rtol=1e-13)
    assert np.all(out[0] == 0) and np.all(out[-1] == 0) and np.all(out[:, 0] == 0) and np.all(out[:, -1] == 0)
    ramp4 = np.add.outer(np.arange(4.0), np.arange(4.0))
    outp = oracle.doProlongation(4, ramp4, 8, fill=0.0)
    assert_bits(outp, golden_ops["kat_prolong_4to8"], "KAT prolong golden")
    np.testing.assert_allclose(outp, np.add.outer(ix, ix) * 3.0 / 7.0, rtol=1e-12, atol=1e-13)


def test_oracle_tables_vs_golden(oracle, golden_tables):
    N = 32768
    while N // 2 >= 8:
        M = N // 2
        lo, w = oracle.restriction_table(N, M)
        assert np.array_equal(lo, golden_tables[f"rt_lo_{N}to{M}"])
        assert_bits(w, golden_tables[f"rt_w_{N}to{M}"], "restriction weights")
        assert np.array_equal(oracle.prolongation_owner(M, N), golden_tables[f"po_{M}to{N}"])
        N = M


@pytest.mark.parametrize("name", ["test.txt", "Vcycle.txt", "Wcycle.txt", "VcycleTrigger.txt", "Vcycle128.txt"])
def test_oracle_driver_vs_golden(oracle, golden_reports, golden_e2e, cycle_dir, name):
    res = oracle.run_cycle_file(os.path.join(cycle_dir, name))
    assert res["status"] == 0
    assert res["report"] == golden_reports[name]
    assert res["mg_error"] == pytest.approx(golden_reports[name + ":mg_error"], rel=1e-12)
    for got, want in zip(res["records"], golden_reports[name + ":records"]):
        assert tuple(got[:3]) == tuple(want[:3])
        assert got[3] == pytest.approx(want[3], rel=1e-12, abs=1e-300)
    key = f"final_U_{name}"
    if key in golden_e2e.files:
        assert_bits(res["U"], golden_e2e[key], "final U")


def test_oracle_csv_format(oracle, golden_e2e, tmp_path):
    """doPrint2File (src/MG_solver_CPU.cpp:735-754): the reference program's own CSV for
    test.txt is a committed fixture."""
    out = tmp_path / "mine.csv"
    oracle.print2file(golden_e2e["final_U_test.txt"], str(out))
    assert out.read_text() == open(os.path.join(GOLDEN, "Sol_CPU_test.txt.csv")).read()


def test_surveyed_headline_numbers(golden_reports):
    """SURVEY.md section 4 quotes these final errors from the reference program."""
    assert f"{golden_reports['test.txt:mg_error']:.6f}" == "0.000666"
    assert f"{golden_reports['Vcycle.txt:mg_error']:.6f}" == "0.000876"
    assert f"{golden_reports['Wcycle.txt:mg_error']:.6f}" == "0.000050"
    assert f"{golden_reports['VcycleTrigger.txt:mg_error']:.6f}" == "0.000784"
    assert f"{golden_reports['Vcycle128.txt:mg_error']:.6f}" == "0.000868"
    steps = [r[2] for r in golden_reports["VcycleTrigger.txt:records"] if r[0] != 0]
    assert steps == [2, 2, 2, 4, 14, 2, 2, 2, 2, 2]


def test_pow_is_square():
    """h^2: the oracle, oracle/_ref (-O2) and the engine use dx*dx, which is what an
    optimising build of the reference computes for pow(dx,2).  The shipped Makefile has no
    -O and calls glibc pow(); that agrees for every grid size the shipped cycle files and
    the BASELINE.json configs generate, and differs by one ulp for a few other sizes."""
    import math
    config_sizes = [2 ** k for k in range(3, 16)] + [16, 128, 256, 3, 5, 9, 17, 33, 65]
    for N in config_sizes:
        dx = 1.0 / float(N - 1)
        assert math.pow(dx, 2) == dx * dx
    differing = [N for N in range(3, 6000) if math.pow(1.0 / (N - 1), 2) != (1.0 / (N - 1)) * (1.0 / (N - 1))]
    assert differing == [2948, 3504, 5378, 5719, 5895]  # documented in oracle/mg_oracle.c


# --------------------------------------------------------------------------- vs _ref
@pytest.mark.parametrize("N", [3, 4, 5, 8, 15, 16, 17, 33, 64, 100, 129])
def test_oracle_vs_reference_ops(oracle, reference, N):
    rng = np.random.default_rng(N)
    U, F = rng.random((N, N)), rng.random((N, N)) - 0.5
    assert_bits(oracle.getSource(N, 1.5, 0.25, -0.5), reference.getSource(N, 1.5, 0.25, -0.5), "getSource")
    assert_bits(oracle.getAnalytic(N), reference.getAnalytic(N), "getAnalytic")
    assert_bits(oracle.getResidual(N, 1.3, U, F), reference.getResidual(N, 1.3, U, F), "getResidual")
    assert_bits(oracle.doGridAddition(N, U, F), reference.doGridAddition(N, U, F), "doGridAddition")
    for s in (1, 2, 3, 7):
        a, ea = oracle.doSmoothing(N, 1.0, U, F, s)
        b, eb = reference.doSmoothing(N, 1.0, U, F, s)
        assert_bits(a, b, f"doSmoothing {s}")
        assert ea == pytest.approx(eb, rel=1e-12)
    for M in {N // 2, N - 1, (N + 1) // 2, N // 2 + 1}:
        if M >= 3:
            assert_bits(oracle.doRestriction(N, U, M), reference.doRestriction(N, U, M), f"doRestriction {N}->{M}")
    for M in (2 * N, 2 * N - 1, N + 1, 2 * N + 1):
        assert_bits(oracle.doProlongation(N, U, M, fill=7.0), reference.doProlongation(N, U, M, fill=7.0), f"doProlongation {N}->{M}")
    if 4 <= N <= 33:
        assert_bits(oracle.doExactSolver(N, 1.0, F, 1e-7), reference.doExactSolver(N, 1.0, F, 1e-7), "GaussSeidel")


def test_oracle_driver_vs_reference_program(oracle, reference, cycle_dir, tmp_path):
    """The reference PROGRAM (oracle/_ref/MG_CPU_ref) on the shipped cycle files: same
    printed report, same CSV, and the reference operators under the oracle driver give the
    same U bit for bit."""
    # the four shipped files + the two step-0 files of SURVEY.md section 8 row D6 (tests/_cycles.py:EXTRA)
    for name in ["test.txt", "Vcycle.txt", "VcycleTrigger.txt", "Wcycle.txt", "StepZero.txt", "StepZeroHalving.txt"]:
        shutil.copy(os.path.join(cycle_dir, name), tmp_path)
        stdout = subprocess.run([_oracle.REF_EXE, "2", name], cwd=tmp_path, capture_output=True, text=True, check=True).stdout
        body = "".join(l for l in stdout.splitlines(keepends=True)
                       if not l.startswith(("OpenMP threads", "Cycle structure file", "Time Used", "Output file name")))
        mine = oracle.run_cycle_file(str(tmp_path / name))
        theirs = oracle.run_cycle_file(str(tmp_path / name), ops=reference)
        assert mine["report"] == body and theirs["report"] == body
        assert_bits(mine["U"], theirs["U"], "final U")
        oracle.print2file(mine["U"], str(tmp_path / "mine.csv"))
        assert (tmp_path / "mine.csv").read_text() == (tmp_path / ("Sol_CPU_" + name)).read_text()


def test_synth_hash_matches_scalar_recipe():
    v = _synth.hash_uniform(5, 3, 11)
    def one(i, seed):
        m = (1 << 64) - 1
        z = ((i + seed) * 0x9E3779B97F4A7C15) & m
        z ^= z >> 30; z = (z * 0xBF58476D1CE4E5B9) & m
        z ^= z >> 27; z = (z * 0x94D049BB133111EB) & m
        z ^= z >> 31
        return (z >> 11) / 9007199254740992.0
    assert [one(i, 11) for i in (5, 6, 7)] == list(v)
    a = np.array([[0.0, -0.0], [1.5, -2.25]])
    s0, s1 = _synth.checksum(a)
    bits = (a + 0.0).view(np.uint64).reshape(-1)
    assert s0 == int(sum(int(b) for b in bits) % (1 << 64))
    assert s1 == int(sum(int(b) * (2 * i + 1) for i, b in enumerate(bits)) % (1 << 64))
