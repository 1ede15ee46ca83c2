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


def text(name):
    a, b, c, nodes = CYCLES[name]
    return "\n".join([a, b, c] + nodes + ["2"])


def write(name, directory):
    path = os.path.join(str(directory), name)
    with open(path, "w") as f:
        f.write(text(name))
    return path


def write_all(directory):
    return {name: write(name, directory) for name in CYCLES}
