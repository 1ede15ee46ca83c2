"""Build the engine IN-TREE with hipcc for gfx950: lib/libmgpoisson.so (kernels + C ABI +
cycle driver) and bin/MG_HIP (the command-line program).  hipcc cross-compiles without a
GPU.  -ffp-contract=off is part of the contract: results are bit-identical to the
reference only without FMA contraction (hard part 4)."""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "lib", "libmgpoisson.so")
EXE = os.path.join(PKG, "bin", "MG_HIP")
SOURCES = ["mg_stream_f32.hip", "mg_stream.hip", "mg_tile.hip", "mg_tile_f32.hip", "mg_kernels.hip", "mg_tail.hip", "mg_tail_f32.hip", "mg_abi.cpp", "mg_tables.cpp", "mg_cycle.cpp", "mg_slab.cpp", "mg_comm.cpp"]
ARCH = "gfx950"


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm; this engine is HIP-only)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def ensure_built():
    """Build the in-tree library when it is absent (a source-only checkout on a box with hipcc).  One
    process at a time: the ranks of a torchrun launch serialise on a lock file."""
    if os.path.exists(LIB) and not (shutil.which("hipcc") and _stale(LIB, _deps())):
        return LIB  # up to date, or a box without hipcc (the prebuilt library is all there is)
    import fcntl
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    with open(os.path.join(os.path.dirname(LIB), ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return build()  # a no-op when another rank built it meanwhile
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _deps():
    deps = _sources() + [os.path.join(CSRC, "mg_internal.h"), os.path.join(ROOT, "include", "mg_hip.h"),
                         os.path.abspath(__file__)]
    return deps + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp", ".inc"))]


def build(force=False, verbose=False):
    srcs = _sources()
    deps = _deps()
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    common = [hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
              "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-Wall", "-Wno-unused-result",
              # the same bytes wherever the tree is checked out (__FILE__ in the error macros): bench.py ties the
              # committed rocprofv3 traffic record to the sha of this library
              f"-ffile-prefix-map={ROOT}=."]
    common += os.environ.get("MG_EXTRA_CXXFLAGS", "").split()
    if force or _stale(LIB, deps):
        objs, jobs = [], []
        for s in srcs:
            o = os.path.join(PKG, "lib", os.path.basename(s) + ".o")
            if force or _stale(o, deps if s.endswith(".hip") else [s] + deps[len(srcs):]):
                # -cuid: hipcc derives the compilation-unit id (part of internal symbol names) from the source PATH by
                # default; a fixed one per file keeps the library's bytes independent of where the tree lies
                cmd = common + (["-x", "hip"] if s.endswith(".hip") else []) + ["-cuid=mg_" + os.path.basename(s).replace(".", "_"), "-c", s, "-o", o]
                if os.path.basename(s).startswith("mg_tail"):
                    # the coarse tail is one workgroup whose exact solver is ONE wave running a 2 KB loop: with the loop
                    # heads on instruction-cache lines the solve measured 10 % faster (the streaming kernels: no change)
                    cmd.insert(-4, "-falign-loops=64")
                jobs.append(cmd)
            objs.append(o)
        # the compilation units are independent: one hipcc per core (the two streaming-kernel units are most of the time)
        from concurrent.futures import ThreadPoolExecutor
        def _run(cmd):
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
        with ThreadPoolExecutor(max_workers=max(1, min(len(jobs) or 1, os.cpu_count() or 1, 8))) as ex:
            list(ex.map(_run, jobs))
        # RCCL is resolved with dlopen on the first communicator call (mg_comm.cpp): no -lrccl here
        cmd = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-o", LIB] + objs + ["-lpthread", "-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    if force or _stale(EXE, [LIB, os.path.join(CSRC, "mg_main.cpp")]):
        cmd = [hipcc(), "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(CSRC, "mg_main.cpp"), "-o", EXE,
               "-L" + os.path.dirname(LIB), "-lmgpoisson", "-Wl,-rpath,$ORIGIN/../lib"]
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
