#!/usr/bin/env python3
"""Generate tests/golden/* from the REFERENCE's own code (oracle/_ref/libmgref.so and
oracle/_ref/MG_CPU_ref, compiled by oracle/Makefile from /root/reference/src where the
sources lie).  Run in the build container only; the fixtures are data (inputs and the
reference's outputs), the reference source never enters this repository.

    python tests/golden/make_golden.py [--big]

--big additionally regenerates golden_fullsize.json (checksums of reference outputs at
N = 4096 .. 32768; needs ~12 GiB of RAM and a few minutes); --add-16384 only adds the
smoothing/residual checksums at N = 16384 to the existing file.
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _oracle as o  # noqa: E402
import _synth  # noqa: E402

REF_SRC = "/root/reference/src"
CYCLES = ["test.txt", "Vcycle.txt", "Wcycle.txt", "VcycleTrigger.txt"]


def lcg_field(N, seed):
    return o.lcg_uniform(N * N, seed).reshape(N, N)


def per_op_vectors(ref):
    """(1) per-operator vectors, SURVEY.md section 8c."""
    out = {}
    for N in (16, 64):
        F_src = ref.getSource(N)
        U_rnd = lcg_field(N, 0x9E3779B97F4A7C15 + N)
        F_rnd = lcg_field(N, 0x1234567 + N)
        out[f"N{N}_F_source"] = F_src
        out[f"N{N}_analytic"] = ref.getAnalytic(N)
        out[f"N{N}_U_rand"] = U_rnd
        out[f"N{N}_F_rand"] = F_rnd
        for tag, U0, F in (("src", np.zeros((N, N)), F_src), ("rnd", U_rnd, F_rnd)):
            for s in (1, 3, 10):
                U, e = ref.doSmoothing(N, 1.0, U0, F, s)
                out[f"N{N}_{tag}_smooth{s}_U"] = U
                out[f"N{N}_{tag}_smooth{s}_err"] = np.array([e])
            U3 = out[f"N{N}_{tag}_smooth3_U"]
            D = ref.getResidual(N, 1.0, U3, F)
            out[f"N{N}_{tag}_residual"] = D
            out[f"N{N}_{tag}_restrict_negD"] = ref.doRestriction(N, -D, N // 2)
            out[f"N{N}_{tag}_prolong_from_half"] = ref.doProlongation(N // 2, U3[: N // 2, : N // 2].copy(), N)
            out[f"N{N}_{tag}_add"] = ref.doGridAddition(N, U3, F)
    # non-nested / odd pairs
    for (Nf, Mc) in ((17, 9), (16, 15), (33, 16)):
        Uf = lcg_field(Nf, 77 + Nf)
        out[f"restrict_{Nf}to{Mc}_in"] = Uf
        out[f"restrict_{Nf}to{Mc}_out"] = ref.doRestriction(Nf, Uf, Mc)
    for (Nc, Mf) in ((15, 16), (9, 17), (16, 33)):
        Uc = lcg_field(Nc, 99 + Nc)
        out[f"prolong_{Nc}to{Mf}_in"] = Uc
        out[f"prolong_{Nc}to{Mf}_out"] = ref.doProlongation(Nc, Uc, Mf, fill=0.0)
    # exact solver
    for N in (8, 16, 17):
        F = lcg_field(N, 4242 + N) - 0.5
        out[f"gs_N{N}_F"] = F
        out[f"gs_N{N}_U"] = ref.doExactSolver(N, 1.0, F, 1e-7)
    # (2) analytic KATs of the reference's own test programs
    # testFunction/Test_doRestriction_GPU.cu:189-193 (N=16 -> 8, Uf = i + j)
    ramp16 = np.add.outer(np.arange(16.0), np.arange(16.0))
    out["kat_restrict_16to8"] = ref.doRestriction(16, ramp16, 8)
    # testFunction/Test_doProlongation_GPU.cu:190-194 (N=4 -> 8, Uc = i + j)
    ramp4 = np.add.outer(np.arange(4.0), np.arange(4.0))
    out["kat_prolong_4to8"] = ref.doProlongation(4, ramp4, 8, fill=0.0)
    return out


def end_to_end(orc, ref):
    """(3) end-to-end goldens: reference program output on the shipped cycle files."""
    import _cycles
    cyc_dir = tempfile.mkdtemp(prefix="mgcycles_")
    _cycles.write_all(cyc_dir)
    # the token streams of tests/_cycles.py ARE the reference's shipped files
    for name in CYCLES:
        ref_tokens = [float(t) for t in open(os.path.join(REF_SRC, name)).read().split()]
        assert ref_tokens == [float(t) for t in _cycles.text(name).split()], name
    arrays, reports = {}, {}
    for name in CYCLES + ["Vcycle128.txt"]:
        d = tempfile.mkdtemp()
        shutil.copy(os.path.join(cyc_dir, name), d)
        stdout = subprocess.run([o.REF_EXE, "4", name], cwd=d, capture_output=True, text=True, check=True).stdout
        keep = [l for l in stdout.splitlines(keepends=True)
                if not l.startswith(("OpenMP threads", "Cycle structure file", "Time Used", "Output file name"))]
        reports[name] = "".join(keep)
        # the CSV is too lossy for a golden (6 decimals); take U from the reference operators
        # run under the (pinned) oracle driver and check the CSV of that U equals the program's
        res = orc.run_cycle_file(os.path.join(d, name), ops=ref)
        assert res["status"] == 0 and res["report"] == reports[name], name
        orc.print2file(res["U"], os.path.join(d, "again.csv"))
        assert open(os.path.join(d, "again.csv")).read() == open(os.path.join(d, "Sol_CPU_" + name)).read()
        if res["N"] <= 256:
            arrays[f"final_U_{name}"] = res["U"]
        reports[name + ":records"] = res["records"]
        reports[name + ":mg_error"] = res["mg_error"]
        if name == "test.txt":
            shutil.copy(os.path.join(d, "Sol_CPU_" + name), os.path.join(HERE, "Sol_CPU_test.txt.csv"))
        shutil.rmtree(d)
    return arrays, reports


def level_tables(orc):
    """(4) R/P index tables 32768 -> 8 (from the pinned restatement; the full-size
    checksums of golden_fullsize.json tie them to the reference itself)."""
    out = {}
    N = 32768
    while N // 2 >= 8:
        M = N // 2
        lo, w = orc.restriction_table(N, M)
        out[f"rt_lo_{N}to{M}"] = lo
        out[f"rt_w_{N}to{M}"] = w
        out[f"po_{M}to{N}"] = orc.prolongation_owner(M, N)
        N = M
    return out


def fullsize(ref, smoothing_sizes=(4096, 8192, 16384), transfer_sizes=(8192, 16384, 32768)):
    """checksums of REFERENCE outputs on hash-generated inputs at benchmark sizes."""
    out = {}
    for N in smoothing_sizes:
        U = _synth.hash_field(N, 11)
        F = _synth.hash_field(N, 22)
        U3, e = ref.doSmoothing(N, 1.0, U, F, 3)
        out[f"smooth3_N{N}"] = {"checksum": _synth.checksum(U3), "error": e}
        out[f"residual_N{N}"] = {"checksum": _synth.checksum(ref.getResidual(N, 1.0, U, F))}
        del U3, F
        print("smoothing/residual", N, flush=True)
    for N in transfer_sizes:
        M = N // 2
        Uf = _synth.hash_field(N, 33)
        out[f"restrict_{N}to{M}"] = {"checksum": _synth.checksum(ref.doRestriction(N, Uf, M))}
        del Uf
        Uc = _synth.hash_field(M, 44)
        out[f"prolong_{M}to{N}"] = {"checksum": _synth.checksum(ref.doProlongation(M, Uc, N, fill=0.0))}
        del Uc
        print("restrict/prolong", N, flush=True)
    return out


def main():
    o.build_oracle()
    orc, ref = o.Oracle(), o.Reference()
    if "--add-16384" in sys.argv:  # only the smoothing/residual checksums at N = 16384 (configs 4/5), merged into the file
        path = os.path.join(HERE, "golden_fullsize.json")
        with open(path) as f:
            have = json.load(f)
        have.update(fullsize(ref, smoothing_sizes=(16384,), transfer_sizes=()))
        with open(path, "w") as f:
            json.dump(have, f, indent=1)
        return
    ops = per_op_vectors(ref)
    e2e_arrays, reports = end_to_end(orc, ref)
    np.savez_compressed(os.path.join(HERE, "golden_ops.npz"), **ops)
    np.savez_compressed(os.path.join(HERE, "golden_e2e.npz"), **e2e_arrays)
    np.savez_compressed(os.path.join(HERE, "golden_tables.npz"), **level_tables(orc))
    with open(os.path.join(HERE, "golden_reports.json"), "w") as f:
        json.dump(reports, f, indent=1)
    if "--big" in sys.argv:
        with open(os.path.join(HERE, "golden_fullsize.json"), "w") as f:
            json.dump(fullsize(ref), f, indent=1)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
