"""The cycle-structure inputs of the tests, written out from their token streams (format:
README.md:43-128 of the reference).  The first four carry the parameters of the reference's
shipped src/test.txt, src/Vcycle.txt, src/Wcycle.txt and src/VcycleTrigger.txt (header `L min_x
min_y / con_step con_N / N_max N_min`, then the node stream); the golden reports under
tests/golden/ were produced by the reference program on exactly these token streams
(tests/golden/make_golden.py checks that against the shipped files).  Vcycle128.txt is
BASELINE.json's config 1."""
import os

V_DOWN_UP = lambda levels, tol: ["-1"] * levels + ["0", f"{tol} 1"] + ["1"] * levels

CYCLES = {
    "test.txt": ("1.0 0.0 0.0", "3 1", "16 8", ["-1", "0", "0.00000001 1", "1"]),
    "Vcycle.txt": ("1.0 0.0 0.0", "3 1", "256 8", V_DOWN_UP(5, "0.0000001")),
    "VcycleTrigger.txt": ("1.0 0.0 0.0", "-1 1", "256 8", V_DOWN_UP(5, "0.0000001")),
    "Wcycle.txt": ("1.0 0.0 0.0", "3 1", "256 8",
                   ["-1", "-1", "-1", "0", "0.00000001 1", "1", "-1", "0", "0.00000001 1", "1", "1",
                    "-1", "-1", "0", "0.00000001 1", "1", "-1", "0", "0.00000001 1", "1", "1", "1"]),
    "Vcycle128.txt": ("1.0 0.0 0.0", "3 1", "128 8", V_DOWN_UP(4, "0.0000001")),
}

# SURVEY.md section 8 row D6 (src/MG_solver_CPU.cpp:241-243, :296-299, :409-411): manual steps (con_step = 0)
# with a step count of 0.  A 0-step `-1` node does nothing at all (no smoothing, no push -- the FMG stub); a
# 0-step `1` node prolongs and adds but does not smooth.  Not one of the shipped files: the expected output
# comes from the reference program run on this token stream (tests/test_oracle_pin.py) and from the oracle.
EXTRA = {
    # con_N = 0: sizes are given per node, so the skipped node does not disturb the size walk
    "StepZero.txt": ("1.0 0.0 0.0", "0 0", "32 8",
                     ["-1", "0 16", "-1", "3 16", "-1", "0 8", "-1", "2 8", "0", "0.0000001 1", "1", "0", "1", "3",
                      "-1", "2 16", "0", "0.000001 1", "1", "0"]),
    # con_N = 1: the reference advances len_flag on the skipped node as well (:176-180), so the next real
    # descent goes 64 -> 16 (N_array[2]) and the way back up walks len_flag down again
    "StepZeroHalving.txt": ("1.0 0.0 0.0", "0 1", "64 8",
                            ["-1", "0", "-1", "3", "0", "0.0000001 1", "1", "0", "-1", "2", "0", "0.000001 1", "1", "2"]),
}
CYCLES_ALL = dict(CYCLES, **EXTRA)


def text(name):
    a, b, c, nodes = CYCLES_ALL[name]
    return "\n".join([a, b, c] + nodes + ["2"])


def write(name, directory):
    path = os.path.join(str(directory), name)
    with open(path, "w") as f:
        f.write(text(name))
    return path


def write_all(directory):
    return {name: write(name, directory) for name in CYCLES_ALL}
