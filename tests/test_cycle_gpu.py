"""End-to-end parity of the cycle-file driver (libmgpoisson.so: mg_cycle_*) with the CPU
oracle driver and the golden outputs of the reference program, on a real MI355X."""
import ctypes as C
import os
import re
import shutil
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, assert_bits

pytestmark = pytest.mark.gpu

CYCLES = ["test.txt", "Vcycle.txt", "Wcycle.txt", "VcycleTrigger.txt", "Vcycle128.txt"]
_num = re.compile(r"^[-+]?\d+\.\d+(e[-+]?\d+)?$")


def reports_match(a, b, tol=2e-6):
    """Same text; numbers printed with %lf may differ in the last printed digit when a
    norm's summation order differs (unspecified even in the reference: OpenMP reduction)."""
    if a == b:
        return True
    ta, tb = a.split(), b.split()
    if len(ta) != len(tb):
        return False
    for x, y in zip(ta, tb):
        if x == y:
            continue
        if _num.match(x) and _num.match(y) and abs(float(x) - float(y)) <= tol:
            continue
        return False
    return True


def check_against(got, want, zero_sign):
    assert got["status"] == 0 and want["status"] == 0
    assert_bits(got["U"], want["U"], "final U", zero_sign=zero_sign)
    assert got["mg_error"] == pytest.approx(want["mg_error"], rel=1e-10)
    assert len(got["records"]) == len(want["records"])
    for g, w in zip(got["records"], want["records"]):
        assert tuple(g[:3]) == tuple(w[:3])
        assert g[3] == pytest.approx(w[3], rel=1e-12, abs=1e-300)
    assert reports_match(got["report"], want["report"])


@pytest.mark.parametrize("name", CYCLES)
@pytest.mark.parametrize("mode", ["unfused", "fused", "graph"])
def test_shipped_cycle_files_vs_oracle_and_golden(mg, oracle, golden_reports, golden_e2e, cycle_dir, name, mode):
    path = os.path.join(cycle_dir, name)
    plan = mg.CyclePlan(path, fused=(mode != "unfused"), graph=(mode == "graph"))
    want = oracle.run_cycle_file(path)
    runs = 3 if mode == "graph" else 1  # warm, capture, replay
    for _ in range(runs):
        got = plan.execute(fetch_U=True)
        check_against(got, want, zero_sign=(mode != "unfused"))
    # golden outputs of the reference program (generated in the build container; F comes from
    # libm exp there and here, so hold the arrays to 1e-12 rather than bitwise)
    assert reports_match(got["report"], golden_reports[name])
    assert got["mg_error"] == pytest.approx(golden_reports[name + ":mg_error"], rel=1e-9)
    key = f"final_U_{name}"
    if key in golden_e2e.files:
        np.testing.assert_allclose(got["U"], golden_e2e[key], rtol=1e-11, atol=1e-15)
    plan.close()


@pytest.mark.parametrize("smoother", ["stream", "simple"])
def test_generated_vcycle_1024_vs_oracle(mg, oracle, tmp_path, smoother):
    mg.set_smoother(smoother)
    try:
        path = str(tmp_path / "V1024.txt")
        assert mg.write_vcycle_file(path, 1024, 8, 3, 1e-7) == 8
        want = oracle.run_cycle_file(path)
        for fused, graph in ((False, False), (True, False), (True, True)):
            plan = mg.CyclePlan(path, fused=fused, graph=graph)
            for _ in range(3 if graph else 1):
                got = plan.execute(fetch_U=True)
            check_against(got, want, zero_sign=fused)
            assert got["graph_replayed"] == graph and got["schedule_launches"] == 0   # (a V-cycle has nothing to merge)
            plan.close()
    finally:
        mg.set_smoother("stream")


@pytest.mark.parametrize("steps", [1, 2, 4])
def test_other_sweep_counts_through_the_cycle_driver(mg, oracle, tmp_path, steps):
    """V(1,1), V(2,2), V(4,4) and the W-cycle of the same sweep counts: the recomputing node pair (levels from
    MG_RECOMPUTE_MIN_N on: 256 in this suite) is instantiated for every pair of at most 6 sweeps together, 4+4 falls back
    to store/re-read (its pipeline of 8 levels does not fit) -- all bit for bit like the oracle, eager and replayed from a
    graph."""
    assert mg.lib().mg_recompute_pair_available(steps, steps) == (1 if steps <= 3 else 0)
    for kind, N in (("V", 1024), ("W", 512)):
        path = str(tmp_path / f"{kind}{steps}.txt")
        (mg.write_vcycle_file if kind == "V" else mg.write_wcycle_file)(path, N, 8, steps, 1e-7)
        want = oracle.run_cycle_file(path)
        for graph in (False, True):
            plan = mg.CyclePlan(path, fused=True, graph=graph)
            for _ in range(3 if graph else 1):
                got = plan.execute(fetch_U=True)
            check_against(got, want, zero_sign=True)
            plan.close()


@pytest.mark.parametrize("down,up", [(1, 2), (2, 1), (1, 3), (3, 1), (2, 3), (3, 2), (1, 4), (4, 1), (2, 4), (4, 2), (4, 3), (3, 4)])
def test_unequal_sweep_counts_per_node(mg, oracle, tmp_path, down, up):
    """Cycle files with per-node step counts (con_step = 0, src/MG_solver_CPU.cpp:171-189, :331-344): `down` sweeps before
    every restriction, `up` after every prolongation.  The driver looks the matching `1` node's count up when it leaves a
    level and drops that level's U whenever the recomputing pair exists for the two counts (down + up <= 6:
    mg_recompute_pair_available; 4+3 and 3+4 stay store/re-read) -- V- and W-shaped files, bit for bit like the oracle."""
    assert mg.lib().mg_recompute_pair_available(down, up) == (1 if down + up <= 6 else 0)
    for kind, N in (("V", 1024), ("W", 512)):
        sizes = []
        n = N
        while n >= 8:
            sizes.append(n)
            n //= 2
        last = len(sizes) - 1

        def visit(level):
            if level == last:
                return ["0", "0.0000001 1"]
            body = ["-1", str(down)] + visit(level + 1) + ["1", str(up)]
            return body * (2 if kind == "W" and level > 0 else 1)

        path = str(tmp_path / f"{kind}_{down}_{up}.txt")
        with open(path, "w") as f:
            f.write(f"1.0 0.0 0.0\n0 1\n{N} 8\n" + "\n".join(visit(0)) + "\n2")
        want = oracle.run_cycle_file(path)
        plan = mg.CyclePlan(path, fused=True)
        for _ in range(2):
            check_against(plan.execute(fetch_U=True), want, zero_sign=True)
        plan.close()


def test_generated_wcycle_full_depth_vs_oracle(mg, oracle, tmp_path):
    """config 3 shape (W-cycle recursion down to N=8) at a size the oracle finishes fast."""
    path = str(tmp_path / "W256.txt")
    assert mg.write_wcycle_file(path, 256, 8, 3, 1e-7) == 6
    want = oracle.run_cycle_file(path)
    plan = mg.CyclePlan(path, fused=True)
    check_against(plan.execute(fetch_U=True), want, zero_sign=True)
    # the 8 x 8 solves run inside the coarse-tail kernel, pipelined over its waves (sweeper + judges): same iterates,
    # same stopping iteration as the reference's `while (err > target_error)` (the last solve's count is left behind)
    assert mg.lastExactSolverIterations() == oracle.gs_iterations()
    plan.close()


@pytest.mark.parametrize("N,n_min,tol", [(64, 8, 1e-7), (64, 8, 1e-3), (64, 8, 1e-9), (48, 6, 1e-7), (32, 4, 1e-8), (128, 8, 3e-6)])
def test_pipelined_exact_solver_in_the_tail(mg, oracle, tmp_path, N, n_min, tol):
    """The in-tail exact solver for even coarsest grids of at most 64 points (8 x 8, 6 x 6, 4 x 4) at several targets
    (few iterations, many iterations): final U bit for bit and the iteration count against the oracle."""
    path = str(tmp_path / "v.txt")
    mg.write_vcycle_file(path, N, n_min, 2, tol)
    want = oracle.run_cycle_file(path)
    plan = mg.CyclePlan(path, fused=True)
    for _ in range(2):
        check_against(plan.execute(fetch_U=True), want, zero_sign=True)
        assert mg.lastExactSolverIterations() == oracle.gs_iterations()
    plan.close()


def test_wcycle_2048_full_depth_vs_oracle(mg, oracle, tmp_path):
    """BASELINE.json config 3 shape (W-cycle, recursion down to N = 8, 2^k coarse solves) at the
    largest size the oracle still finishes in seconds; exercises the coarse-tail kernel with
    a long node slice (36 nodes per visit of level 64)."""
    path = str(tmp_path / "W2048.txt")
    assert mg.write_wcycle_file(path, 2048, 8, 3, 1e-7) == 9
    want = oracle.run_cycle_file(path, want_report=False)
    for graph in (False, True):
        plan = mg.CyclePlan(path, fused=True, graph=graph, report=False)
        for _ in range(3 if graph else 1):
            got = plan.execute(fetch_U=True)
        assert got["status"] == 0
        assert_bits(got["U"], want["U"], "W-cycle 2048 final U", zero_sign=True)
        assert len(got["records"]) == len(want["records"])
        for g, w in zip(got["records"], want["records"]):
            assert tuple(g[:3]) == tuple(w[:3]) and g[3] == pytest.approx(w[3], rel=1e-12, abs=1e-300)
        plan.close()


def test_back_to_back_windows(mg, oracle, tmp_path):
    """mg_cycle_enqueue x K + mg_cycle_collect == K independent runs (each starts from U = 0)."""
    path = str(tmp_path / "V512.txt")
    mg.write_vcycle_file(path, 512, 8, 3, 1e-7)
    want = oracle.run_cycle_file(path)
    for graph in (False, True):
        plan = mg.CyclePlan(path, fused=True, graph=graph)
        plan.execute()
        if graph:
            plan.execute()
        for _ in range(5):
            plan.enqueue()
        got = plan.collect(fetch_U=True)
        check_against(got, want, zero_sign=True)
        plan.close()


@pytest.mark.parametrize("N,steps", [(1024, 3), (512, 2)])
def test_wcycle_batched_schedule_back_to_back(mg, oracle, tmp_path, N, steps):
    """W-cycles run as a batched breadth-first schedule: the second descent from a level reads that level's F and nothing
    else (the reference zeroes U before every pre-smoothing, src/MG_solver_CPU.cpp:252-257), so all visits of a level are
    independent and ONE launch carries them all (mg_cycle.cpp: mg_cycle_plan::sched).  Several windows back to back, eager
    and replayed from a graph (the schedule is single-stream: it is captured like a V-cycle): every window's final U,
    every record and the last coarse solve's iteration count as the oracle's."""
    path = str(tmp_path / "W.txt")
    mg.write_wcycle_file(path, N, 8, steps, 1e-7)
    want = oracle.run_cycle_file(path)
    for graph in (False, True):
        plan = mg.CyclePlan(path, fused=True, graph=graph)
        for _ in range(2):
            check_against(plan.execute(fetch_U=True), want, zero_sign=True)
        for _ in range(4):
            plan.enqueue()
        got = plan.collect(fetch_U=True)
        check_against(got, want, zero_sign=True)
        # the result says how the window ran: as a batched schedule (2 launches per level above the coarse tail + the tail),
        # and -- with MG_CYCLE_GRAPH -- as a replay of the captured graph
        assert got["graph_replayed"] == graph and 0 < got["schedule_launches"] <= 2 * 8 + 1
        assert mg.lastExactSolverIterations() == oracle.gs_iterations()
        plan.close()


@pytest.mark.parametrize("N,n_min,steps,batched", [(704, 8, 3, True), (1152, 9, 3, True), (1448, 8, 2, False), (360, 5, 1, False)])
def test_wcycle_on_hierarchies_that_are_not_powers_of_two(mg, oracle, tmp_path, N, n_min, steps, batched):
    """W-cycles down 704 -> 11, 1152 -> 9 (every level above the coarsest even: batched schedules with the register-tile
    kernel on levels such as 44, 36 and 72 and coarse tails from 22 and 18 with an 11 x 11 / 9 x 9 solve) and down
    1448 -> 11, 360 -> 5 (an odd level in mid-hierarchy -- 181, 45 -- is not a fused node: the trace gives up, the file
    runs node by node): final U bit for bit, every record, the last solve's iteration count."""
    path = str(tmp_path / f"W{N}.txt")
    mg.write_wcycle_file(path, N, n_min, steps, 1e-7)
    want = oracle.run_cycle_file(path)
    plan = mg.CyclePlan(path, fused=True)
    for _ in range(2):
        got = plan.execute(fetch_U=True)
        check_against(got, want, zero_sign=True)
        assert (got["schedule_launches"] > 0) == batched
        assert mg.lastExactSolverIterations() == oracle.gs_iterations()
    plan.close()


def test_wcycle_16384_batched_vs_oracle(mg, oracle, tmp_path):
    """The W-cycle one size above BASELINE config 3 (N = 16384^2: batches of up to 256 instances, 19 launches for 5118
    nodes) against the oracle's own run: the final U through the 128-bit checksum, every record."""
    import _synth
    N = 16384
    path = str(tmp_path / "W16384.txt")
    mg.write_wcycle_file(path, N, 8, 3, 1e-7)
    want = oracle.run_cycle_file(path, want_report=False)
    plan = mg.CyclePlan(path, fused=True, report=False)
    got = plan.execute()
    s = (C.c_uint64 * 2)()
    mg.lib().mg_checksum(got["U_ptr"], N * N, s)
    assert got["status"] == 0 and want["status"] == 0 and 0 < got["schedule_launches"] <= 2 * 11 + 1
    assert (int(s[0]), int(s[1])) == tuple(_synth.checksum(want["U"])), "W-cycle at 16384: final U differs from the oracle's"
    assert len(got["records"]) == len(want["records"])
    for g, w in zip(got["records"], want["records"]):
        assert tuple(g[:3]) == tuple(w[:3]) and g[3] == pytest.approx(w[3], rel=1e-12, abs=1e-300)
    assert got["mg_error"] == pytest.approx(want["mg_error"], rel=1e-10)
    plan.close()
    mg.lib().mg_pool_trim()


def test_batched_plan_follows_a_later_smoother_setting(mg, oracle, tmp_path):
    """A plan whose first window built a batched schedule holds the kernels of the smoother setting it was traced under;
    mg_set_smoother afterwards makes the same plan run node by node with the other kernels, and back: every window the
    oracle's, and the result says which way it ran."""
    path = str(tmp_path / "W512.txt")
    mg.write_wcycle_file(path, 512, 8, 3, 1e-7)
    want = oracle.run_cycle_file(path)
    plan = mg.CyclePlan(path, fused=True)
    try:
        got = plan.execute(fetch_U=True)
        check_against(got, want, zero_sign=True)
        assert got["schedule_launches"] > 0
        mg.set_smoother("simple")
        got = plan.execute(fetch_U=True)
        check_against(got, want, zero_sign=True)
        assert got["schedule_launches"] == 0
        mg.set_smoother("stream")
        got = plan.execute(fetch_U=True)
        check_against(got, want, zero_sign=True)
        assert got["schedule_launches"] > 0
    finally:
        mg.set_smoother("stream")
        plan.close()


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_irregular_cycle_trees_batched_vs_oracle(mg, oracle, tmp_path, seed):
    """The dataflow trace on cycle files that are neither V nor W: every visit of a level descends 1, 2 or 3 times (drawn
    per visit), so the levels carry batches of irregular sizes and sub-cycles of different lengths sit side by side --
    the schedule's depth analysis has to order every node behind exactly what it reads.  Final U bit for bit, every
    record, the report and the last solve's iteration count as the oracle's; eager and replayed from a graph."""
    rng = np.random.default_rng(seed)
    N = 512 if seed % 2 else 256
    sizes = []
    n = N
    while n >= 8:
        sizes.append(n)
        n //= 2
    last = len(sizes) - 1

    def visit(level):
        if level == last:
            return ["0", "0.0000001 1"]
        out = []
        for _ in range(int(rng.integers(1, 4)) if level > 0 else 1):
            out += ["-1"] + visit(level + 1) + ["1"]
        return out

    path = str(tmp_path / f"tree{seed}.txt")
    with open(path, "w") as f:
        f.write(f"1.0 0.0 0.0\n2 1\n{N} 8\n" + "\n".join(visit(0)) + "\n2")
    want = oracle.run_cycle_file(path)
    for graph in (False, True):
        plan = mg.CyclePlan(path, fused=True, graph=graph)
        for _ in range(3 if graph else 2):
            got = plan.execute(fetch_U=True)
            check_against(got, want, zero_sign=True)
        assert got["graph_replayed"] == graph
        assert mg.lastExactSolverIterations() == oracle.gs_iterations()
        plan.close()


def test_wcycle_with_a_standalone_coarse_solve_is_interpreted(mg, oracle, tmp_path):
    """A W-cycle whose coarsest level lies above the LDS tail (512 -> 256 -> 128, two exact solves at N = 128 through the
    stand-alone solver and its device-side state): the dataflow trace gives up on the first such node (build_schedule:
    not one fused launch), the plan runs node by node -- and is the oracle's bit for bit like every other file."""
    path = str(tmp_path / "W512_128.txt")
    assert mg.write_wcycle_file(path, 512, 128, 3, 1e-3) == 3
    want = oracle.run_cycle_file(path)
    plan = mg.CyclePlan(path, fused=True)
    for _ in range(2):
        check_against(plan.execute(fetch_U=True), want, zero_sign=True)
    assert mg.lastExactSolverIterations() == oracle.gs_iterations()
    plan.close()


def test_coarse_tail_matches_node_by_node(mg, tmp_path):
    """MG_NO_TAIL=1 (read once per process) cannot be toggled here, so compare the fused driver
    (tail kernel for N <= 64) with the unfused one (operator by operator) on a deep hierarchy."""
    path = str(tmp_path / "V256_4.txt")
    mg.write_vcycle_file(path, 256, 4, 2, 1e-8)  # levels 256 ... 4: the tail holds 64, 32, 16, 8, 4
    a = mg.CyclePlan(path, fused=True)
    b = mg.CyclePlan(path, fused=False)
    ra, rb = a.execute(fetch_U=True), b.execute(fetch_U=True)
    assert ra["status"] == 0 and rb["status"] == 0
    assert_bits(ra["U"], rb["U"], "fused+tail vs unfused", zero_sign=True)
    assert reports_match(ra["report"], rb["report"])
    a.close(); b.close()


def test_manual_grammar_con_step0_con_N0_and_minus_one(mg, oracle, tmp_path):
    """README.md:103-128: con_step=0 / con_N=0 (manual steps and sizes, non-nested sizes)
    and con_N=2 (N-1 per level)."""
    a = tmp_path / "manual.txt"
    a.write_text("1.0 0.0 0.0\n0 0\n33 1\n-1\n2 17\n-1\n4 9\n0\n0.0000001 1\n1\n3\n1\n1\n2")
    b = tmp_path / "minus1.txt"
    b.write_text("1.0 0.0 0.0\n2 2\n20 17\n-1\n-1\n-1\n0\n0.000001 1\n1\n1\n1\n2")
    c = tmp_path / "twocycles.txt"  # two V-cycles chained: the second keeps U (restart rule :252-257)
    c.write_text("1.0 0.0 0.0\n3 1\n64 8\n" + ("-1\n-1\n-1\n0\n0.0000001 1\n1\n1\n1\n" * 2) + "2")
    for f in (a, b, c):
        want = oracle.run_cycle_file(str(f))
        for fused in (False, True):
            plan = mg.CyclePlan(str(f), fused=fused)
            check_against(plan.execute(fetch_U=True), want, zero_sign=fused)
            plan.close()


def test_non_nested_pairs_above_the_recompute_threshold(mg, oracle, tmp_path):
    """Round-2 advisor finding: with con_step = 3 a `-1` node on a level above MG_RECOMPUTE_MIN_N (256 in this suite,
    conftest.py) whose coarse size is not ~N/2 -- manual sizes (con_N = 0: 300 -> 220) or con_N = 2 (N -> N - 1) -- must
    not take the recomputing node pair: its restriction is not a fused stage (samples less than two fine columns
    apart).  The predicate now covers both nodes of the pair; such levels run store/re-read, operator by operator,
    bit for bit like the oracle."""
    a = tmp_path / "manual300.txt"
    a.write_text("1.0 0.0 0.0\n3 0\n300 1\n-1\n220\n-1\n8\n0\n0.0000001 1\n1\n1\n2")
    b = tmp_path / "minus1_258.txt"
    b.write_text("1.0 0.0 0.0\n3 2\n258 255\n-1\n-1\n-1\n0\n0.05 1\n1\n1\n1\n2")
    for f in (a, b):
        want = oracle.run_cycle_file(str(f))
        assert want["status"] == 0
        for fused, graph in ((False, False), (True, False)):
            plan = mg.CyclePlan(str(f), fused=fused, graph=graph)
            check_against(plan.execute(fetch_U=True), want, zero_sign=fused)
            plan.close()


@pytest.mark.parametrize("name", ["StepZero.txt", "StepZeroHalving.txt"])
def test_step_zero_nodes_vs_oracle(mg, oracle, cycle_dir, name):
    """SURVEY.md section 8 row D6 (src/MG_solver_CPU.cpp:241-243, :296-299, :409-411): a 0-step `-1` node does
    nothing (no smoothing, no push; with con_N = 1 it still advances the size walk), a 0-step `1` node prolongs
    and adds without smoothing.  The oracle's handling is pinned to the reference program on these two files in
    tests/test_oracle_pin.py."""
    path = os.path.join(cycle_dir, name)
    want = oracle.run_cycle_file(path)
    assert want["status"] == 0
    for fused, graph in ((False, False), (True, False), (True, True)):
        plan = mg.CyclePlan(path, fused=fused, graph=graph)
        for _ in range(3 if graph else 1):
            got = plan.execute(fetch_U=True)
        check_against(got, want, zero_sign=fused)
        plan.close()


def test_malformed_cycle_files(mg, oracle, tmp_path):
    # more -1 nodes than generated sizes: the reference reads out of bounds;
    # both drivers report status 4 instead
    bad = tmp_path / "toomany.txt"
    bad.write_text("1.0 0.0 0.0\n3 1\n32 8\n-1\n-1\n-1\n-1\n0\n0.0000001 1\n1\n1\n1\n1\n2")
    assert oracle.run_cycle_file(str(bad))["status"] == 4
    plan = mg.CyclePlan(str(bad))
    assert plan.execute()["status"] == 4
    plan.close()
    with pytest.raises(mg.MGError, match="Cannot open file"):
        mg.CyclePlan(str(tmp_path / "does_not_exist.txt"))


def test_command_line_program(mg, cycle_dir, tmp_path):
    """MG_HIP keeps the reference's command line and output format
    (src/MG_solver_CPU.cpp:51-68, :448-459); the CSV of test.txt must equal the one the
    reference program wrote (fixture tests/golden/Sol_CPU_test.txt.csv)."""
    shutil.copy(os.path.join(cycle_dir, "test.txt"), tmp_path)
    out = subprocess.run([mg.EXE_PATH, "4", "test.txt"], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "OpenMP threads = 4" in out.stdout and "Cycle structure file name = test.txt" in out.stdout
    assert "    Error = 0.000666" in out.stdout and "Output file name = Sol_HIP_test.txt" in out.stdout
    assert (tmp_path / "Sol_HIP_test.txt").read_text() == open(os.path.join(GOLDEN, "Sol_CPU_test.txt.csv")).read()


@pytest.mark.parametrize("name", ["test.txt", "Vcycle.txt", "Wcycle.txt", "VcycleTrigger.txt"])
def test_reference_main_drives_the_engine(mg, golden_reports, cycle_dir, tmp_path, name):
    """Drop-in proof: oracle/_ref/MG_HIP_dropin is the REFERENCE's main() and linked list
    (src/MG_solver_CPU.cpp:36-462, src/linkedlist.cpp), compiled where the sources lie with the
    edits of INTEGRATION.md (oracle/dropin_build.sh) against libmgpoisson.so through
    include/mg_dropin.hpp.  It must print what the reference program printed and write the
    same CSV."""
    exe = os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref", "MG_HIP_dropin")
    exe = os.path.abspath(exe)
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/MG_HIP_dropin not built (needs /root/reference at build time)")
    shutil.copy(os.path.join(cycle_dir, name), tmp_path)
    out = subprocess.run([exe, "4", name], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    body = "".join(l for l in out.stdout.splitlines(keepends=True)
                   if not l.startswith(("OpenMP threads", "Cycle structure file", "Time Used", "Output file name")))
    assert reports_match(body, golden_reports[name])
    if name == "test.txt":
        assert (tmp_path / "Sol_CPU_test.txt").read_text() == open(os.path.join(GOLDEN, "Sol_CPU_test.txt.csv")).read()


@pytest.mark.parametrize("kind", ["V", "W"])
def test_headline_size_cycles_vs_oracle(mg, oracle, tmp_path, kind):
    """BASELINE.json's headline size, N = 8192^2 (configs[1] shape at the bench size, configs[2]: the
    W-cycle recursion down to N = 8 with its 512 coarse solves): the fused driver against the oracle's
    run of the same file -- the final U through the 128-bit checksum (computed on the device, so the
    512 MiB array never crosses PCIe), every smoothing error, the analytic error."""
    import _synth
    N = 8192
    path = str(tmp_path / f"{kind}{N}.txt")
    (mg.write_vcycle_file if kind == "V" else mg.write_wcycle_file)(path, N, 8, 3, 1e-7)
    want = oracle.run_cycle_file(path, want_report=False)
    assert want["status"] == 0
    want_sum = _synth.checksum(want["U"])
    plan = mg.CyclePlan(path, fused=True, report=False)
    got = plan.execute()
    assert got["status"] == 0
    out = (C.c_uint64 * 2)()
    mg.lib().mg_checksum(got["U_ptr"], N * N, out)
    assert (int(out[0]), int(out[1])) == want_sum, f"{kind}-cycle at N = {N}: final U differs from the oracle's"
    assert got["mg_error"] == pytest.approx(want["mg_error"], rel=1e-10)
    assert len(got["records"]) == len(want["records"])
    for g, w in zip(got["records"], want["records"]):
        assert tuple(g[:3]) == tuple(w[:3]) and g[3] == pytest.approx(w[3], rel=1e-12, abs=1e-300)
    plan.close()


def test_product_thresholds_against_the_oracle_and_against_the_test_thresholds(mg, oracle, tmp_path):
    """The suite runs the large-grid code paths (recomputing node pair, non-temporal stores) from small sizes on
    (conftest.py).  Here the library runs with ITS OWN thresholds in a child process -- the configuration bench.py
    measures: recomputing pair from N = 4096, store/re-read below -- at N = 8192 against the oracle (checksum of the
    final U, analytic error, every smoothing error) and at N = 16384 against this process's run of the same file."""
    import _synth
    env = {k: v for k, v in os.environ.items() if k not in ("MG_RECOMPUTE_MIN_N", "MG_NT_MIN_N", "MG_F32_COLS4_MIN_N")}
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "_defaults_worker.py"), "8192", "16384"], env=env,
                         capture_output=True, text=True, timeout=600)
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("DEFAULTS_WORKER ")]
    assert out.returncode == 0 and line, out.stdout[-2000:] + out.stderr[-3000:]
    import json
    child = json.loads(line[0][len("DEFAULTS_WORKER "):])
    path = str(tmp_path / "V8192.txt")
    mg.write_vcycle_file(path, 8192, 8, 3, 1e-7)
    want = oracle.run_cycle_file(path, want_report=False)
    assert want["status"] == 0 and child["8192"]["status"] == 0
    assert tuple(child["8192"]["sum"]) == _synth.checksum(want["U"]), "product thresholds: final U at N = 8192 differs from the oracle's"
    assert child["8192"]["mg_error"] == pytest.approx(want["mg_error"], rel=1e-10)
    for g, w in zip(child["8192"]["errors"], want["records"]):
        assert g == pytest.approx(w[3], rel=1e-12, abs=1e-300)
    N = 16384
    path = str(tmp_path / "V16384.txt")
    mg.write_vcycle_file(path, N, 8, 3, 1e-7)
    plan = mg.CyclePlan(path, fused=True, report=False)
    got = plan.execute()
    s = (C.c_uint64 * 2)()
    mg.lib().mg_checksum(got["U_ptr"], N * N, s)
    assert got["status"] == 0 and child["16384"]["status"] == 0
    assert [int(s[0]), int(s[1])] == child["16384"]["sum"], "N = 16384: the two threshold configurations disagree"
    plan.close()


def test_product_thresholds_on_the_small_and_medium_levels(mg, oracle, tmp_path):
    """The same child process (the library's own thresholds: register-tile kernel on the levels 65...1024, streaming
    kernel in its store/re-read form on 2048, recomputing pair from 4096 on) at sizes the oracle finishes in seconds:
    V- and W-cycles with 1...4 sweeps per node, a hierarchy that is not a power of two, and the fp32 cycle of the mixed
    mode against the numpy restatement -- the final U through the 128-bit checksum, every smoothing error."""
    import json
    import _synth
    import _oracle_f32 as o32
    specs = ["V:1024:3", "V:2048:3", "W:512:3", "V:1024:1", "V:1024:2", "V:1024:4", "W:256:2", "V:1448:3", "V:704:3", "MV:1024:3", "MV:512:2"]
    env = {k: v for k, v in os.environ.items() if k not in ("MG_RECOMPUTE_MIN_N", "MG_NT_MIN_N", "MG_F32_COLS4_MIN_N")}
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "_defaults_worker.py")] + specs, env=env,
                         capture_output=True, text=True, timeout=900)
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("DEFAULTS_WORKER ")]
    assert out.returncode == 0 and line, out.stdout[-2000:] + out.stderr[-3000:]
    child = json.loads(line[0][len("DEFAULTS_WORKER "):])
    # the W-cycles once more node by node, in file order (MG_CYCLE_BATCH=0: the interpreter instead of the batched
    # schedule): the same numbers
    serial = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "_defaults_worker.py")] + [sp for sp in specs if sp[0] == "W"],
                            env=dict(env, MG_CYCLE_BATCH="0"), capture_output=True, text=True, timeout=900)
    sline = [ln for ln in serial.stdout.splitlines() if ln.startswith("DEFAULTS_WORKER ")]
    assert serial.returncode == 0 and sline, serial.stdout[-2000:] + serial.stderr[-3000:]
    for sp, rec in json.loads(sline[0][len("DEFAULTS_WORKER "):]).items():
        # (arrays bit for bit; the scalar norms to rounding: the batched schedule hands level 64 to the tile kernel, the
        # interpreter to the coarse-tail kernel, which sum their partials in different orders)
        assert rec["sum"] == child[sp]["sum"], f"{sp}: batched and node-by-node W-cycle differ"
        assert rec["errors"] == pytest.approx(child[sp]["errors"], rel=1e-12, abs=1e-300), f"{sp}: batched and node-by-node W-cycle differ"
    for spec in specs:
        kind, N, steps = spec.split(":")
        N, steps = int(N), int(steps)
        path = str(tmp_path / f"{kind}{N}_{steps}.txt")
        (mg.write_wcycle_file if kind == "W" else mg.write_vcycle_file)(path, N, 8, steps, 1e-7)
        got = child[spec]
        assert got["status"] == 0, spec
        if kind == "MV":
            toks = open(path).read().split()
            sizes, n = [], N
            while n >= 8:
                sizes.append(n)
                n //= 2
            U32, recs = o32.run_cycle_tokens(mg, oracle, oracle.getSource(N), 1.0, steps, sizes, toks[7:])
            assert tuple(got["sum"]) == _synth.checksum(U32.astype(np.float64)), f"{spec}: fp32 result differs from the numpy restatement"
            for g, w in zip(got["errors"], recs):
                assert g == pytest.approx(w[2], rel=1e-10, abs=1e-300), spec
            continue
        want = oracle.run_cycle_file(path, want_report=False)
        assert want["status"] == 0
        assert tuple(got["sum"]) == _synth.checksum(want["U"]), f"{spec}: final U differs from the oracle's"
        assert got["mg_error"] == pytest.approx(want["mg_error"], rel=1e-10)
        assert len(got["errors"]) == len(want["records"])
        for g, w in zip(got["errors"], want["records"]):
            assert g == pytest.approx(w[3], rel=1e-12, abs=1e-300), spec


@pytest.mark.parametrize("kind,N,n_min", [("V", 88, 8), ("V", 1448, 8), ("W", 176, 8), ("V", 104, 8), ("V", 256, 16), ("V", 120, 8)])
def test_cycles_whose_coarsest_level_has_65_to_256_points(mg, oracle, tmp_path, kind, N, n_min):
    """Hierarchies that do not end on an 8 x 8 grid (the weak-scaling grids 11520 and 23040 end on 11 x 11;
    104 -> 13, 256 with N_min 16 -> 16, 120 -> 15): the coarse-tail kernel solves them with the
    one-wave LDS Gauss-Seidel; iterates, iteration count and everything downstream as the oracle's."""
    path = str(tmp_path / "c.txt")
    (mg.write_vcycle_file if kind == "V" else mg.write_wcycle_file)(path, N, n_min, 3, 1e-7)
    want = oracle.run_cycle_file(path)
    plan = mg.CyclePlan(path, fused=True)
    got = plan.execute(fetch_U=True)
    check_against(got, want, zero_sign=True)
    assert mg.lastExactSolverIterations() == oracle.gs_iterations()
    plan.close()


def test_w_cycle_revisits_are_idempotent_in_the_reference_semantics(mg, oracle, tmp_path):
    """Observation recorded in DESIGN.md section 8: the reference zeroes U on every descent (:252-257), so the
    repeated visits of its W-cycle recompute identical values and the W-cycle ends on the V-cycle's U, bit for
    bit -- in the oracle and in the engine alike (the engine executes every visit, as the reference does)."""
    v, w = str(tmp_path / "V.txt"), str(tmp_path / "W.txt")
    mg.write_vcycle_file(v, 256, 8, 3, 1e-7)
    mg.write_wcycle_file(w, 256, 8, 3, 1e-7)
    ov, ow = oracle.run_cycle_file(v), oracle.run_cycle_file(w)
    assert np.array_equal(ov["U"], ow["U"]) and len(ow["records"]) > len(ov["records"])
    gv = mg.CyclePlan(v, fused=True).execute(fetch_U=True)
    gw = mg.CyclePlan(w, fused=True).execute(fetch_U=True)
    assert np.array_equal(gv["U"], gw["U"])
    assert len(gw["records"]) == len(ow["records"])
