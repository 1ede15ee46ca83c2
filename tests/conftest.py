import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

GOLDEN = os.path.join(HERE, "golden")

# The oracle is OpenMP code.  A GPU box exposes far more hardware threads than the CPU
# share a job gets (16 per GPU): an unbounded team spins in every barrier of the tiny
# coarse-grid loops and a 100 ms solve takes minutes.  Bound the team before libgomp loads.
os.environ.setdefault("OMP_NUM_THREADS", str(min(8, os.cpu_count() or 1)))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
# slab plans fill freshly allocated level arrays with NaN: a halo row that nobody wrote (no exchange, no
# redundant computation) then shows up in the compared result instead of passing on stale data
os.environ.setdefault("MG_SLAB_POISON", "1")
# the 4-columns-per-lane form of the fp32 kernels is the product's choice from N = 8192 on (where it pays); the tests
# take it from N = 1024 on, so that the numpy restatement can check it bit for bit at sizes it finishes in seconds
os.environ.setdefault("MG_F32_COLS4_MIN_N", "1024")
# likewise the instantiation of the fused `1` node with non-temporal stores (the product: N >= 8192)
os.environ.setdefault("MG_NT_MIN_N", "1024")
# and the fused node pair that recomputes the pre-smoothed field instead of storing and re-reading it (the product:
# N >= 4096): from N = 256 on, so that the cycle tests against the oracle run through it at every size they use
os.environ.setdefault("MG_RECOMPUTE_MIN_N", "256")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def bits_equal(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


def bits_equal_mod_zero_sign(a, b):
    """bit equality after canonicalising -0.0 to +0.0 (folded sign flip)."""
    return bits_equal(np.asarray(a) + 0.0, np.asarray(b) + 0.0)


def assert_bits(a, b, what="", zero_sign=False):
    ok = bits_equal_mod_zero_sign(a, b) if zero_sign else bits_equal(a, b)
    if not ok:
        a = np.asarray(a); b = np.asarray(b)
        bad = np.argwhere(a.view(np.uint64) != b.view(np.uint64)) if a.shape == b.shape else []
        diff = np.max(np.abs(a - b)) if a.shape == b.shape else None
        raise AssertionError(f"{what}: arrays differ bitwise at {len(bad)} points, first {bad[:4].tolist()}, max|diff|={diff}")


@pytest.fixture(scope="session")
def cycle_dir(tmp_path_factory):
    """Directory holding the cycle-structure inputs of the tests (written from tests/_cycles.py)."""
    import _cycles
    d = tmp_path_factory.mktemp("cycles")
    _cycles.write_all(d)
    return str(d)


@pytest.fixture(scope="session")
def oracle():
    import _oracle
    return _oracle.Oracle()


@pytest.fixture(scope="session")
def reference():
    import _oracle
    if not _oracle.have_reference():
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    return _oracle.Reference()


@pytest.fixture(scope="session")
def golden_ops():
    return np.load(os.path.join(GOLDEN, "golden_ops.npz"))


@pytest.fixture(scope="session")
def golden_e2e():
    return np.load(os.path.join(GOLDEN, "golden_e2e.npz"))


@pytest.fixture(scope="session")
def golden_tables():
    return np.load(os.path.join(GOLDEN, "golden_tables.npz"))


@pytest.fixture(scope="session")
def golden_reports():
    with open(os.path.join(GOLDEN, "golden_reports.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_fullsize():
    with open(os.path.join(GOLDEN, "golden_fullsize.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def mg():
    """The engine, initialised on cuda:0.  Fails (does not skip) when the HIP library or
    the device is missing: GPU tests must never pass on a fallback."""
    from multigrid_poisson_solver_amd import build as b
    b.ensure_built()   # source-only checkout: compile the in-tree library first (hipcc); never a fallback
    import multigrid_poisson_solver_amd as m
    m.init(0)
    yield m
    m.finalize()
