"""ctypes bindings for the CPU oracle (oracle/libmgoracle.so) and, when built, the
reference's own operators (oracle/_ref/libmgref.so).  TEST INFRASTRUCTURE: imported
only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libmgoracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libmgref.so")
REF_SO_MAKEFLAGS = os.path.join(ORACLE_DIR, "_ref", "libmgref_makeflags.so")  # src/Makefile:8 flags (no -O)
REF_EXE = os.path.join(ORACLE_DIR, "_ref", "MG_CPU_ref")

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build_oracle():
    """(Re)build the oracle with its Makefile; also builds oracle/_ref when
    /root/reference is mounted."""
    subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True)


class NodeRecord(C.Structure):
    _fields_ = [("node", C.c_int), ("N", C.c_int), ("steps", C.c_int), ("error", C.c_double)]


class Result(C.Structure):
    _fields_ = [("N", C.c_int), ("U", C.POINTER(C.c_double)), ("mg_error", C.c_double),
                ("time_ms", C.c_double), ("n_records", C.c_int),
                ("records", C.POINTER(NodeRecord)), ("status", C.c_int)]


class Ops(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("getSource", "getAnalytic", "getResidual", "doGridAddition",
                                          "doSmoothing", "doExactSolver", "doRestriction",
                                          "doProlongation")]


_SIGS = {
    "getSource": [C.c_int, C.c_double, _dp, C.c_double, C.c_double],
    "getAnalytic": [C.c_int, C.c_double, _dp, C.c_double, C.c_double],
    "getResidual": [C.c_int, C.c_double, _dp, _dp, _dp],
    "doGridAddition": [C.c_int, _dp, _dp],
    "doSmoothing": [C.c_int, C.c_double, _dp, _dp, C.c_int, C.POINTER(C.c_double)],
    "doExactSolver": [C.c_int, C.c_double, _dp, _dp, C.c_double, C.c_int],
    "doRestriction": [C.c_int, _dp, C.c_int, _dp],
    "doProlongation": [C.c_int, _dp, C.c_int, _dp],
}


class _Operators:
    """Same call surface as the reference (src/MG_solver_CPU.cpp:23-28) on numpy arrays."""

    def __init__(self, lib, prefix):
        self.lib, self.prefix = lib, prefix
        for name, sig in _SIGS.items():
            fn = getattr(lib, prefix + name)
            fn.argtypes, fn.restype = sig, None

    def _f(self, name):
        return getattr(self.lib, self.prefix + name)

    def getSource(self, N, L=1.0, min_x=0.0, min_y=0.0):
        F = np.empty((N, N)); self._f("getSource")(N, L, F, min_x, min_y); return F

    def getAnalytic(self, N, L=1.0, min_x=0.0, min_y=0.0):
        U = np.empty((N, N)); self._f("getAnalytic")(N, L, U, min_x, min_y); return U

    def getResidual(self, N, L, U, F):
        D = np.empty((N, N)); self._f("getResidual")(N, L, np.ascontiguousarray(U), np.ascontiguousarray(F), D); return D

    def doGridAddition(self, N, U1, U2):
        out = np.array(U1, dtype=np.float64, copy=True, order="C")
        self._f("doGridAddition")(N, out, np.ascontiguousarray(U2)); return out

    def doSmoothing(self, N, L, U, F, step):
        out = np.array(U, dtype=np.float64, copy=True, order="C")
        err = C.c_double(0.0)
        self._f("doSmoothing")(N, L, out, np.ascontiguousarray(F), step, C.byref(err))
        return out, err.value

    def doExactSolver(self, N, L, F, tol, option=1):
        U = np.full((N, N), np.nan)
        self._f("doExactSolver")(N, L, U, np.ascontiguousarray(F), tol, option); return U

    def doRestriction(self, N, U_f, M):
        out = np.full((M, M), np.nan)
        self._f("doRestriction")(N, np.ascontiguousarray(U_f), M, out); return out

    def doProlongation(self, N, U_c, M, fill=np.nan):
        out = np.full((M, M), fill)
        self._f("doProlongation")(N, np.ascontiguousarray(U_c), M, out); return out


class Oracle(_Operators):
    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        lib = C.CDLL(ORACLE_SO)
        super().__init__(lib, "orc_")
        lib.orc_restrictionTable.argtypes = [C.c_int, C.c_int, _ip, _dp]
        lib.orc_prolongationOwner.argtypes = [C.c_int, C.c_int, _ip]
        lib.orc_runCycleFile.argtypes = [C.c_char_p, C.POINTER(Ops), C.POINTER(Result), C.POINTER(C.c_char_p)]
        lib.orc_runCycleFile.restype = C.c_int
        lib.orc_free.argtypes = [C.c_void_p]
        lib.orc_print2File.argtypes = [C.c_int, _dp, C.c_char_p]
        lib.orc_lastExactSolverIterations.restype = C.c_int
        lib.orc_maxThreads.restype = C.c_int

    def set_threads(self, n):
        self.lib.orc_setThreads(int(n))

    def max_threads(self):
        return self.lib.orc_maxThreads()

    def gs_iterations(self):
        return self.lib.orc_lastExactSolverIterations()

    def restriction_table(self, N, M):
        lo = np.empty(M, dtype=np.int32); w = np.empty(M)
        self.lib.orc_restrictionTable(N, M, lo, w); return lo, w

    def prolongation_owner(self, N, M):
        o = np.empty(M, dtype=np.int32); self.lib.orc_prolongationOwner(N, M, o); return o

    def run_cycle_file(self, path, ops=None, want_report=True):
        """Run the oracle driver.  ops: None (oracle operators) or a Reference instance."""
        res = Result()
        rep = C.c_char_p()
        ops_struct = None
        if ops is not None:
            ops_struct = Ops(*[C.cast(getattr(ops.lib, ops.prefix + n), C.c_void_p) for n, _ in Ops._fields_])
        # keep the char* as a raw pointer so it can be freed
        rep_raw = C.c_void_p()
        status = self.lib.orc_runCycleFile(
            os.fsencode(path), C.byref(ops_struct) if ops_struct else None, C.byref(res),
            C.cast(C.byref(rep_raw), C.POINTER(C.c_char_p)) if want_report else None)
        N = res.N
        U = np.ctypeslib.as_array(res.U, shape=(N, N)).copy() if N else None
        records = [(res.records[i].node, res.records[i].N, res.records[i].steps, res.records[i].error)
                   for i in range(res.n_records)]
        text = C.string_at(rep_raw).decode() if (want_report and rep_raw.value) else ""
        self.lib.orc_free(C.cast(res.U, C.c_void_p)); self.lib.orc_free(C.cast(res.records, C.c_void_p))
        if rep_raw.value:
            self.lib.orc_free(rep_raw)
        return dict(status=status, N=N, U=U, mg_error=res.mg_error, time_ms=res.time_ms,
                    records=records, report=text)

    def print2file(self, U, path):
        N = U.shape[0]
        return self.lib.orc_print2File(N, np.ascontiguousarray(U), os.fsencode(path))


class Reference(_Operators):
    """The reference's own operators, compiled from /root/reference by oracle/Makefile."""

    def __init__(self, makefile_flags=False):
        lib = C.CDLL(REF_SO_MAKEFLAGS if makefile_flags else REF_SO)
        super().__init__(lib, "ref_")

    def set_threads(self, n):
        self.lib.ref_setThreads(int(n))


def have_reference():
    return os.path.exists(REF_SO)


def lcg_uniform(n, seed=0x9E3779B97F4A7C15):
    """Seeded 64-bit LCG -> uniform [0,1) doubles (not rand())."""
    out = np.empty(n, dtype=np.float64)
    s = seed & 0xFFFFFFFFFFFFFFFF
    for i in range(n):
        s = (s * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        out[i] = (s >> 11) * (1.0 / 9007199254740992.0)
    return out
