"""multigrid_poisson_solver_amd -- Python harness over the C ABI of libmgpoisson.so.

The product is the HIP library (csrc/, built by build.py) behind include/mg_hip.h; this
module only binds it with ctypes for the tests and the benchmark.  It mirrors the
reference's operator interface (src/MG_solver_CPU.cpp:23-28: getResidual,
doGridAddition, doSmoothing, doExactSolver, doRestriction, doProlongation) on
device-resident arrays.  There is no CPU fallback: importing works anywhere, but the
first call raises if the library or a HIP device is missing.
"""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MG_LIB") or os.path.join(_PKG, "lib", "libmgpoisson.so")  # MG_LIB: A/B builds
EXE_PATH = os.path.join(_PKG, "bin", "MG_HIP")

MG_CYCLE_FUSED, MG_CYCLE_GRAPH, MG_CYCLE_REPORT, MG_CYCLE_ERROR, MG_CYCLE_MIXED = 1, 2, 4, 8, 16


class MGError(RuntimeError):
    pass


class NodeRecord(C.Structure):
    _fields_ = [("node", C.c_int), ("N", C.c_int), ("steps", C.c_int), ("error", C.c_double)]


class ProfileEntry(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("N", C.c_int), ("launches", C.c_int), ("total_ms", C.c_double),
                ("algo_bytes", C.c_double)]


class CycleResult(C.Structure):
    _fields_ = [("status", C.c_int), ("N", C.c_int), ("U_dev", C.c_void_p), ("mg_error", C.c_double),
                ("time_ms", C.c_double), ("device_ms", C.c_double), ("n_records", C.c_int),
                ("records", C.POINTER(NodeRecord)), ("report", C.c_char_p), ("graph_replayed", C.c_int),
                ("schedule_launches", C.c_int)]


_vp, _i, _d, _sz, _u64 = C.c_void_p, C.c_int, C.c_double, C.c_size_t, C.c_uint64
_dptr = C.POINTER(C.c_double)

# every symbol include/mg_hip.h declares: name -> (restype, argtypes)
ABI = {
    "mg_init": (_i, [_i]), "mg_finalize": (None, []), "mg_set_stream": (None, [_vp]),
    "mg_get_stream": (_vp, []), "mg_sync": (None, []), "mg_last_error": (_i, []),
    "mg_last_error_string": (C.c_char_p, []), "mg_clear_error": (None, []),
    "mg_set_abort_on_error": (None, [_i]), "mg_set_smoother": (_i, [C.c_char_p]),
    "mg_version": (C.c_char_p, []), "mg_set_source": (_i, [C.c_char_p]), "mg_source_mode": (C.c_char_p, []),
    "mg_source_is_bit_identical": (_i, []),
    "mg_alloc": (_vp, [_sz]), "mg_free": (None, [_vp]), "mg_pool_trim": (None, []),
    "mg_pool_bytes": (_sz, []), "mg_upload": (None, [_vp, _vp, _sz]), "mg_download": (None, [_vp, _vp, _sz]),
    "mg_copy": (None, [_vp, _vp, _sz]), "mg_fill_zero": (None, [_vp, _sz]), "mg_negate": (None, [_i, _vp]),
    "mg_getSource": (None, [_i, _d, _vp, _d, _d]), "mg_getAnalytic": (None, [_i, _d, _vp, _d, _d]),
    "mg_analyticError": (None, [_i, _d, _vp, _d, _d, _dptr]),
    "mg_getResidual": (None, [_i, _d, _vp, _vp, _vp]), "mg_doGridAddition": (None, [_i, _vp, _vp]),
    "mg_doSmoothing": (None, [_i, _d, _vp, _vp, _i, _dptr]),
    "mg_doExactSolver": (None, [_i, _d, _vp, _vp, _d, _i]),
    "mg_doRestriction": (None, [_i, _vp, _i, _vp]), "mg_doProlongation": (None, [_i, _vp, _i, _vp]),
    "mg_smooth_pp": (None, [_i, _d, _vp, _vp, _vp, _i, _vp, _vp, _i]),
    "mg_smooth_restrict": (None, [_i, _d, _vp, _vp, _vp, _i, _vp, _i, _vp]),
    "mg_prolong_smooth": (None, [_i, _vp, _i, _d, _vp, _vp, _vp, _i, _vp]),
    "mg_cycle_set_refinement": (_i, [_vp, _i]), "mg_cycle_refinement_errors": (_i, [_vp, _vp, _i]),
    "mg_smooth_restrict_f32": (None, [_i, _d, _vp, _vp, _vp, _i, _vp, _i, _vp]),
    "mg_prolong_smooth_f32": (None, [_i, _vp, _i, _d, _vp, _vp, _vp, _i, _vp]),
    "mg_alloc_f32": (_vp, [_sz]), "mg_free_f32": (None, [_vp]), "mg_to_f32": (None, [_vp, _vp, _sz]),
    "mg_to_f64": (None, [_vp, _vp, _sz]), "mg_upload_f32": (None, [_vp, _vp, _sz]),
    "mg_download_f32": (None, [_vp, _vp, _sz]),
    "mg_prolongAdd": (None, [_i, _vp, _i, _vp, _vp]), "mg_restrict_signed": (None, [_i, _vp, _i, _vp, _i]),
    "mg_lastExactSolverIterations": (_i, []),
    "mg_restriction_table": (None, [_i, _i, _vp, _vp]),
    "mg_prolongation_table": (None, [_i, _i, _i, _vp, _vp, _vp]),
    "mg_fill_uniform": (None, [_vp, _sz, _u64]), "mg_checksum": (None, [_vp, _sz, C.POINTER(_u64)]),
    "mg_cycle_load": (_vp, [C.c_char_p, _i]), "mg_cycle_execute": (_i, [_vp, C.POINTER(CycleResult)]),
    "mg_cycle_enqueue": (_i, [_vp]), "mg_cycle_collect": (_i, [_vp, C.POINTER(CycleResult)]),
    "mg_cycle_destroy": (None, [_vp]), "mg_cycle_main": (_i, [_i, C.POINTER(C.c_char_p)]),
    "mg_print2File": (_i, [_i, _vp, C.c_char_p]),
    "mg_comm_unique_id_bytes": (_i, []), "mg_comm_get_unique_id": (_i, [_vp]),
    "mg_comm_init_host": (_i, [_i, _i, _vp]),
    "mg_comm_init": (_i, [_i, _i, _vp]), "mg_comm_finalize": (None, []), "mg_comm_selftest": (_i, [_sz]), "mg_comm_rank": (_i, []),
    "mg_comm_size": (_i, []),
    "mg_comm_library": (C.c_char_p, []),
    "mg_slab_partition": (_i, [_i, _i, _i, _i, _vp, _vp]), "mg_slab_ghost_rows": (_i, []),
    "mg_slab_set_refinement": (_i, [_vp, _i]), "mg_slab_refinement_errors": (_i, [_vp, _vp, _i]),
    "mg_slab_schedule": (_i, [_i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "mg_slab_recompute_levels": (_i, [_i, _i, _i, _vp]), "mg_slab_recompute_levels_ranks": (_i, [_i, _i, _i, _i, _vp]),
    "mg_recompute_pair_available": (_i, [_i, _i]),
    "mg_slab_load": (_vp, [C.c_char_p, _i, _i, _i]), "mg_slab_load_flags": (_vp, [C.c_char_p, _i, _i, _i, _i]), "mg_slab_execute": (_i, [_vp, C.POINTER(CycleResult)]),
    "mg_slab_enqueue": (_i, [_vp]), "mg_slab_collect": (_i, [_vp, C.POINTER(CycleResult)]),
    "mg_slab_gather_U": (_i, [_vp, _vp]), "mg_slab_want_error": (None, [_vp, _i]), "mg_slab_destroy": (None, [_vp]),
    "mg_profile_begin": (None, [_i]), "mg_profile_sample": (None, [_i]), "mg_profile_end": (_i, [C.POINTER(ProfileEntry), _i]),
}

_lib = None
hip_runtime = None   # which libamdhip64 the engine was bound to ("system", or the path of torch's copy)


def _bind_hip_runtime():
    """ONE HIP runtime per process.  libmgpoisson.so needs `libamdhip64.so.7`; a PyTorch wheel bundles its own
    copy (soname libamdhip64.so.7, but referenced by torch under the name `libamdhip64.so`, so the loader does
    not recognise a copy mapped earlier from /opt/rocm and maps a second one -- two runtimes on one device abort
    at exit).  The other order is fine: once torch's copy is mapped the engine's NEEDED entry matches it by
    soname.  So: when torch is installed but not imported yet, map ITS runtime first; every later import order
    then ends up with that single copy.  MG_HIP_RUNTIME=system keeps the ROCm installation's runtime (for
    processes that never import torch), MG_HIP_RUNTIME=<path> names a library explicitly."""
    global hip_runtime
    mode = os.environ.get("MG_HIP_RUNTIME", "auto")
    hip_runtime = "system"
    if mode == "system":
        return
    if mode not in ("auto", "torch"):
        C.CDLL(mode, mode=C.RTLD_GLOBAL)
        hip_runtime = mode
        return
    if "torch" in sys.modules:
        hip_runtime = "torch (imported before the engine)"
        return
    try:
        spec = importlib.util.find_spec("torch")  # locates the package without importing it
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)
        hip_runtime = cand


def load_library(path=None):
    """dlopen libmgpoisson.so and type every exported symbol.  Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise MGError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(hipcc, gfx950).  There is no CPU fallback.")
    _bind_hip_runtime()
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    missing = []
    for name, (res, args) in ABI.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype, fn.argtypes = res, args
    if missing:
        raise MGError(f"{path} does not export: {missing}")
    _lib = lib
    return lib


def _check():
    code = _lib.mg_last_error()
    if code:
        msg = _lib.mg_last_error_string().decode()
        _lib.mg_clear_error()
        raise MGError(f"[{code}] {msg}")


_initialised = False


def init(device=0):
    global _initialised
    lib = load_library()
    lib.mg_set_abort_on_error(0)
    if lib.mg_init(int(device)) != 0:
        code = lib.mg_last_error()
        msg = lib.mg_last_error_string().decode()
        lib.mg_clear_error()
        raise MGError(f"mg_init({device}) failed [{code}]: {msg}")
    _initialised = True
    return lib


def lib():
    if not _initialised:
        init(int(os.environ.get("LOCAL_RANK", "0")) if os.environ.get("MG_DEVICE") is None
             else int(os.environ["MG_DEVICE"]))
    return _lib


def finalize():
    global _initialised
    if _initialised:
        _lib.mg_finalize()
        _initialised = False


def sync():
    lib().mg_sync()
    _check()


def set_smoother(name):
    lib().mg_set_smoother(name.encode())
    _check()


def set_source(mode):
    lib().mg_set_source(mode.encode())
    _check()


def source_mode():
    """'host' or 'device': where getSource is evaluated (auto mode: the device when it reproduces the host's libm)."""
    m = lib().mg_source_mode().decode()
    _check()
    return m


class DeviceGrid:
    """A device-resident fp64 array obtained from mg_alloc (the engine's replacement for
    the malloc'ed U/F/D of src/linkedlist.cpp:9-11)."""

    def __init__(self, shape):
        if isinstance(shape, int):
            shape = (shape, shape)
        self.shape = tuple(int(s) for s in shape)
        self.size = int(np.prod(self.shape))
        self.ptr = lib().mg_alloc(self.size)
        _check()
        if not self.ptr:
            raise MGError("mg_alloc returned NULL")

    @property
    def N(self):
        return self.shape[0]

    @classmethod
    def from_host(cls, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        g = cls(a.shape)
        _lib.mg_upload(g.ptr, a.ctypes.data, a.size)
        _check()
        return g

    @classmethod
    def zeros(cls, shape):
        g = cls(shape)
        _lib.mg_fill_zero(g.ptr, g.size)
        return g

    @classmethod
    def uniform(cls, shape, seed):
        g = cls(shape)
        _lib.mg_fill_uniform(g.ptr, g.size, seed)
        return g

    def to_host(self):
        out = np.empty(self.shape, dtype=np.float64)
        _lib.mg_download(out.ctypes.data, self.ptr, self.size)
        _check()
        return out

    def copy(self):
        g = DeviceGrid(self.shape)
        _lib.mg_copy(g.ptr, self.ptr, self.size)
        return g

    def checksum(self):
        out = (C.c_uint64 * 2)()
        _lib.mg_checksum(self.ptr, self.size, out)
        _check()
        return int(out[0]), int(out[1])

    def free(self):
        if self.ptr and _initialised:
            _lib.mg_free(self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceGrid32:
    """fp32 device array (mixed-precision mode)."""

    def __init__(self, shape):
        if isinstance(shape, int):
            shape = (shape, shape)
        self.shape = tuple(int(s) for s in shape)
        self.size = int(np.prod(self.shape))
        self.ptr = lib().mg_alloc_f32(self.size)
        _check()

    @classmethod
    def from_host(cls, a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        g = cls(a.shape)
        _lib.mg_upload_f32(g.ptr, a.ctypes.data, a.size)
        _check()
        return g

    def to_host(self):
        out = np.empty(self.shape, dtype=np.float32)
        _lib.mg_download_f32(out.ctypes.data, self.ptr, self.size)
        _check()
        return out

    def free(self):
        if self.ptr and _initialised:
            _lib.mg_free_f32(self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def smooth_restrict_f32(N, L, U_out, F, step, M, F_c, want_error=False):
    err = DeviceGrid((1,)) if want_error else None
    lib().mg_smooth_restrict_f32(N, L, None, U_out.ptr, F.ptr, step, err.ptr if err else None, M, F_c.ptr)
    _check()
    if want_error:
        v = float(err.to_host()[0])
        err.free()
        return v


def prolong_smooth_f32(Nc, U_c, N, L, U_in, U_out, F, step, want_error=False):
    err = DeviceGrid((1,)) if want_error else None
    lib().mg_prolong_smooth_f32(Nc, U_c.ptr, N, L, U_in.ptr, U_out.ptr, F.ptr, step, err.ptr if err else None)
    _check()
    if want_error:
        v = float(err.to_host()[0])
        err.free()
        return v


# ---------------------------------------------------------------------------------
# the reference's operator surface (src/MG_solver_CPU.cpp:23-28) on DeviceGrid
# ---------------------------------------------------------------------------------
def getSource(N, L=1.0, min_x=0.0, min_y=0.0):
    F = DeviceGrid(N)
    _lib.mg_getSource(N, L, F.ptr, min_x, min_y)
    _check()
    return F


def getAnalytic(N, L=1.0, min_x=0.0, min_y=0.0):
    U = DeviceGrid(N)
    _lib.mg_getAnalytic(N, L, U.ptr, min_x, min_y)
    _check()
    return U


def analyticError(N, L, U, min_x=0.0, min_y=0.0):
    e = C.c_double()
    lib().mg_analyticError(N, L, U.ptr, min_x, min_y, C.byref(e))
    _check()
    return e.value


def getResidual(N, L, U, F, D):
    lib().mg_getResidual(N, L, U.ptr, F.ptr, D.ptr)
    _check()


def doGridAddition(N, U1, U2):
    lib().mg_doGridAddition(N, U1.ptr, U2.ptr)
    _check()


def doSmoothing(N, L, U, F, step, want_error=True):
    """In place, like the reference; returns the error scalar (host double)."""
    e = C.c_double()
    lib().mg_doSmoothing(N, L, U.ptr, F.ptr, step, C.byref(e) if want_error else None)
    _check()
    return e.value if want_error else None


def doExactSolver(N, L, U, F, target_error, option=1):
    lib().mg_doExactSolver(N, L, U.ptr, F.ptr, target_error, option)
    _check()


def lastExactSolverIterations():
    n = lib().mg_lastExactSolverIterations()
    _check()
    return n


def doRestriction(N, U_f, M, U_c):
    lib().mg_doRestriction(N, U_f.ptr, M, U_c.ptr)
    _check()


def doProlongation(N, U_c, M, U_f):
    lib().mg_doProlongation(N, U_c.ptr, M, U_f.ptr)
    _check()


def negate(N, D):
    lib().mg_negate(N, D.ptr)
    _check()


def smooth_pp(N, L, U_in, U_out, F, step, want_error=False, D_out=None, d_sign=1):
    """Out-of-place fused form (mg_smooth_pp).  U_in None = all-zero start."""
    err = DeviceGrid((1,)) if want_error else None
    lib().mg_smooth_pp(N, L, U_in.ptr if U_in is not None else None, U_out.ptr, F.ptr, step,
                       err.ptr if err else None, D_out.ptr if D_out is not None else None, d_sign)
    _check()
    if want_error:
        v = float(err.to_host()[0])
        err.free()
        return v
    return None


def smooth_restrict(N, L, U_in, U_out, F, step, M, F_c, want_error=False):
    err = DeviceGrid((1,)) if want_error else None
    lib().mg_smooth_restrict(N, L, U_in.ptr if U_in is not None else None, U_out.ptr, F.ptr, step,
                             err.ptr if err else None, M, F_c.ptr)
    _check()
    if want_error:
        v = float(err.to_host()[0])
        err.free()
        return v


def prolong_smooth(Nc, U_c, N, L, U_in, U_out, F, step, want_error=False):
    err = DeviceGrid((1,)) if want_error else None
    lib().mg_prolong_smooth(Nc, U_c.ptr, N, L, U_in.ptr, U_out.ptr, F.ptr, step, err.ptr if err else None)
    _check()
    if want_error:
        v = float(err.to_host()[0])
        err.free()
        return v


def prolongAdd(N, U_c, M, U_f_in, U_f_out):
    lib().mg_prolongAdd(N, U_c.ptr, M, U_f_in.ptr, U_f_out.ptr)
    _check()


def restrict_signed(N, U_f, M, U_c, sign):
    lib().mg_restrict_signed(N, U_f.ptr, M, U_c.ptr, sign)
    _check()


def restriction_table(N, M):
    lo = np.empty(M, dtype=np.int32)
    w = np.empty(M, dtype=np.float64)
    load_library().mg_restriction_table(N, M, lo.ctypes.data, w.ctypes.data)
    return lo, w


def prolongation_table(N, M, axis):
    owner = np.empty(M, dtype=np.int32)
    hi = np.empty(M)
    lo = np.empty(M)
    load_library().mg_prolongation_table(N, M, axis, owner.ctypes.data, hi.ctypes.data, lo.ctypes.data)
    return owner, hi, lo


def profile_begin(min_N=0, every=1):
    lib().mg_profile_sample(int(every))
    lib().mg_profile_begin(int(min_N))
    _check()


def profile_end(cap=256):
    buf = (ProfileEntry * cap)()
    n = lib().mg_profile_end(buf, cap)
    _check()
    return [dict(name=buf[i].name.decode(), N=buf[i].N, launches=buf[i].launches, total_ms=buf[i].total_ms,
                 algo_bytes=buf[i].algo_bytes) for i in range(n)]


class CyclePlan:
    """mg_cycle_load / mg_cycle_execute: the reference program's timed window
    (src/MG_solver_CPU.cpp:156..429) over a cycle structure file."""

    def __init__(self, path, fused=True, graph=False, report=True, error=True, mixed=False, refinement=1):
        flags = ((MG_CYCLE_FUSED if fused else 0) | (MG_CYCLE_GRAPH if graph else 0) |
                 (MG_CYCLE_REPORT if report else 0) | (MG_CYCLE_ERROR if error else 0) |
                 (MG_CYCLE_MIXED if mixed else 0))
        try:
            with open(path) as f:
                head = f.read().split()[:3]
        except OSError as e:
            raise MGError(f"Cannot open file {path}") from e  # src/MG_solver_CPU.cpp:65-68
        self.L, self.min_x, self.min_y = (float(t) for t in head)
        self._plan = lib().mg_cycle_load(os.fsencode(path), flags)
        _check()
        if not self._plan:
            raise MGError(f"cannot load cycle file {path}")
        self.refinement = refinement
        if refinement != 1:
            _lib.mg_cycle_set_refinement(self._plan, refinement)
            _check()

    def enqueue(self):
        """One window on the engine's stream, no host synchronisation (see collect)."""
        status = _lib.mg_cycle_enqueue(self._plan)
        _check()
        return status

    def collect(self, fetch_U=False):
        return self.execute(fetch_U=fetch_U, _collect_only=True)

    def execute(self, fetch_U=False, _collect_only=False):
        res = CycleResult()
        status = (_lib.mg_cycle_collect if _collect_only else _lib.mg_cycle_execute)(self._plan, C.byref(res))
        _check()
        out = dict(status=status, N=res.N, mg_error=res.mg_error, time_ms=res.time_ms, device_ms=res.device_ms,
                   records=[(res.records[i].node, res.records[i].N, res.records[i].steps, res.records[i].error)
                            for i in range(res.n_records)],
                   report=res.report.decode() if res.report else "", U_ptr=res.U_dev,
                   graph_replayed=bool(res.graph_replayed), schedule_launches=res.schedule_launches)
        if fetch_U:
            U = np.empty((res.N, res.N))
            _lib.mg_download(U.ctypes.data, res.U_dev, U.size)
            out["U"] = U
        if self.refinement > 1:
            e = np.zeros(self.refinement - 1)
            n = _lib.mg_cycle_refinement_errors(self._plan, e.ctypes.data, e.size)
            out["refinement_errors"] = e[:n].tolist()
        return out

    def analytic_error(self, result):
        """sum|analytic - U|/N^2 of a finished execute (src/MG_solver_CPU.cpp:434-445)."""
        e = C.c_double()
        _lib.mg_analyticError(result["N"], self.L, result["U_ptr"], self.min_x, self.min_y, C.byref(e))
        _check()
        return e.value

    def close(self):
        if self._plan and _initialised:
            _lib.mg_cycle_destroy(self._plan)
        self._plan = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def slab_partition(N_max, N_min, nranks, collapse_N):
    """Host-only: [(level N, collapsed, [(lo, hi) per rank])] of the row-slab decomposition."""
    lib_ = load_library()
    nl = lib_.mg_slab_partition(N_max, N_min, nranks, collapse_N, None, None)
    out = np.zeros((nl, nranks, 2), dtype=np.int32)
    coll = np.zeros(nl, dtype=np.int32)
    lib_.mg_slab_partition(N_max, N_min, nranks, collapse_N, out.ctypes.data, coll.ctypes.data)
    sizes, n = [], N_max
    while n >= N_min and n > 0:
        sizes.append(n)
        n //= 2
    return [(sizes[l], bool(coll[l]), [tuple(int(v) for v in out[l, r]) for r in range(nranks)]) for l in range(nl)]


def slab_schedule(N_max, N_min, nranks, collapse_N, steps, ca_mode=-1, ca_pct=-1):
    """mg_slab_schedule (host-only): per level a dict N, collapsed, halo, needF, xF, xU, pre (> 0: the level's `1`
    launch recomputes the pre-smoothed field, its `-1` launch does not store it) and, per rank, the row ranges
    own / dext / ext / fwr as (lo, hi) tuples."""
    lib_ = load_library()
    nl = lib_.mg_slab_partition(N_max, N_min, nranks, collapse_N, None, None)
    lev = np.zeros((nl, 6), dtype=np.int32)
    rk = np.zeros((nl, nranks, 8), dtype=np.int32)
    n = lib_.mg_slab_schedule(N_max, N_min, nranks, collapse_N, steps, ca_mode, ca_pct, lev.ctypes.data, rk.ctypes.data)
    if n < 0:
        lib_.mg_clear_error()
        raise MGError("mg_slab_schedule: a halo does not fit the neighbouring slab (raise collapse_N)")
    pre = np.zeros(nl, dtype=np.int32)
    lib_.mg_slab_recompute_levels_ranks(N_max, N_min, steps, nranks, pre.ctypes.data)
    out = []
    for l in range(nl):
        d = dict(N=int(lev[l, 0]), collapsed=bool(lev[l, 1]), halo=int(lev[l, 2]), needF=int(lev[l, 3]), xF=int(lev[l, 4]),
                 xU=int(lev[l, 5]), pre=0 if lev[l, 1] else int(pre[l]))
        for k, name in enumerate(("own", "dext", "ext", "fwr")):
            d[name] = [(int(rk[l, r, 2 * k]), int(rk[l, r, 2 * k + 1])) for r in range(nranks)]
        out.append(d)
    return out


def slab_ghost_rows():
    return load_library().mg_slab_ghost_rows()


def comm_init(rank, nranks, unique_id_bytes):
    buf = C.create_string_buffer(bytes(unique_id_bytes), len(unique_id_bytes))
    if lib().mg_comm_init(rank, nranks, buf) != 0:
        _check()
        raise MGError("mg_comm_init failed")
    _check()


_EXCHANGE_CB = C.CFUNCTYPE(_i, _vp, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_vp), C.POINTER(_sz))
_ALLGATHER_CB = C.CFUNCTYPE(_i, _vp, _vp, _vp, _sz)


class HostTransport(C.Structure):
    _fields_ = [("user", _vp), ("exchange", _EXCHANGE_CB), ("allgather", _ALLGATHER_CB)]


_host_transport_keepalive = []


def comm_init_host(rank, nranks, exchange, allgather):
    """mg_comm_init_host with Python callables:
    exchange(ops) with ops = [(is_send, peer, uint8 ndarray view of the host buffer), ...] must
    complete every transfer before returning; allgather(send, recv) fills recv (nranks x count)."""

    def _exchange(_user, n, is_send, peer, buf, count):
        try:
            ops = [(bool(is_send[i]), int(peer[i]),
                    np.ctypeslib.as_array(C.cast(buf[i], C.POINTER(C.c_uint8)), shape=(int(count[i]),))) for i in range(n)]
            exchange(ops)
            return 0
        except Exception as e:  # never unwind through the C frame
            print(f"host transport exchange failed: {e!r}", flush=True)
            return 1

    def _allgather(_user, send, recv, count):
        try:
            a = np.ctypeslib.as_array(C.cast(send, C.POINTER(_d)), shape=(int(count),))
            b = np.ctypeslib.as_array(C.cast(recv, C.POINTER(_d)), shape=(nranks, int(count)))
            allgather(a, b)
            return 0
        except Exception as e:
            print(f"host transport allgather failed: {e!r}", flush=True)
            return 1

    t = HostTransport(None, _EXCHANGE_CB(_exchange), _ALLGATHER_CB(_allgather))
    _host_transport_keepalive.append(t)
    if lib().mg_comm_init_host(rank, nranks, C.byref(t)) != 0:
        _check()
        raise MGError("mg_comm_init_host failed")
    _check()


def comm_unique_id():
    n = lib().mg_comm_unique_id_bytes()
    buf = C.create_string_buffer(n)
    if _lib.mg_comm_get_unique_id(buf) != 0:
        _check()
        raise MGError("mg_comm_get_unique_id failed")
    return bytes(buf.raw)


class SlabPlan:
    """The cycle-file driver on a 1-D row-slab decomposition (mg_slab_*)."""

    def __init__(self, path, nranks, rank=-1, collapse_N=512, mixed=False, refinement=1):
        self._plan = lib().mg_slab_load_flags(os.fsencode(path), nranks, rank, collapse_N, MG_CYCLE_MIXED if mixed else 0)
        _check()
        if not self._plan:
            raise MGError(f"cannot load cycle file {path} in row-slab mode")
        self.refinement = refinement
        if refinement != 1:
            _lib.mg_slab_set_refinement(self._plan, refinement)
            _check()

    def enqueue(self):
        status = _lib.mg_slab_enqueue(self._plan)
        _check()
        return status

    def collect(self):
        return self.execute(_collect_only=True)

    def execute(self, _collect_only=False):
        res = CycleResult()
        status = (_lib.mg_slab_collect if _collect_only else _lib.mg_slab_execute)(self._plan, C.byref(res))
        _check()
        out = dict(status=status, N=res.N, mg_error=res.mg_error, time_ms=res.time_ms, device_ms=res.device_ms,
                   records=[(res.records[i].node, res.records[i].N, res.records[i].steps, res.records[i].error)
                            for i in range(res.n_records)])
        if self.refinement > 1:
            e = np.zeros(self.refinement - 1)
            n = _lib.mg_slab_refinement_errors(self._plan, e.ctypes.data, e.size)
            out["refinement_errors"] = e[:n].tolist()
        return out

    def want_error(self, on):
        _lib.mg_slab_want_error(self._plan, 1 if on else 0)

    def gather_U(self, N):
        U = np.zeros((N, N))
        _lib.mg_slab_gather_U(self._plan, U.ctypes.data)
        _check()
        return U

    def close(self):
        if self._plan and _initialised:
            _lib.mg_slab_destroy(self._plan)
        self._plan = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def write_vcycle_file(path, N, N_min=8, steps=3, tol=1e-7, L=1.0):
    """The synthetic V-cycle of SURVEY.md section 8d: header `steps 1 / N N_min`, one -1
    per level down to the coarsest generated size, one exact solve, one 1 per level up."""
    levels = 0
    n = N
    while n >= N_min:
        levels += 1
        n //= 2
    with open(path, "w") as f:
        f.write(f"{L} 0.0 0.0\n{steps} 1\n{N} {N_min}\n")
        f.write("-1\n" * (levels - 1))
        f.write(f"0\n{tol:.10f} 1\n")
        f.write("1\n" * (levels - 1))
        f.write("2")
    return levels


def write_wcycle_file(path, N, N_min=8, steps=3, tol=1e-7, L=1.0, depth=None):
    """W-cycle with the recursion of the shipped src/Wcycle.txt (which stops after 4 of
    its 6 generated levels); depth=None takes it down to the coarsest generated size."""
    sizes = []
    n = N
    while n >= N_min:
        sizes.append(n)
        n //= 2
    if depth is not None:
        sizes = sizes[:depth]
    last = len(sizes) - 1

    def visit(level):
        if level == last:
            return ["0", f"{tol:.10f} 1"]
        return ["-1"] + visit(level + 1) + ["1"] + ["-1"] + visit(level + 1) + ["1"]

    nodes = ["-1"] + visit(1) + ["1"] if last >= 1 else visit(0)
    with open(path, "w") as f:
        f.write(f"{L} 0.0 0.0\n{steps} 1\n{N} {N_min}\n" + "\n".join(nodes) + "\n2")
    return len(sizes)
