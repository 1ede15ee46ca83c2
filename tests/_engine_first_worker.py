"""Subprocess worker: the ENGINE is loaded and initialised first, torch is imported afterwards, then the
engine works on torch tensors and the process exits normally (exit code 0 is the assertion: two HIP/RCCL
runtimes in one process used to abort at exit)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
os.environ.setdefault("OMP_NUM_THREADS", "4")
import numpy as np  # noqa: E402
import multigrid_poisson_solver_amd as mg  # noqa: E402

assert "torch" not in sys.modules
mg.init(0)
N = 256
rng = np.random.default_rng(5)
U0, F = rng.random((N, N)), rng.random((N, N)) - 0.5
Ud, Fd = mg.DeviceGrid.from_host(U0), mg.DeviceGrid.from_host(F)
first = mg.DeviceGrid(N)
mg.smooth_pp(N, 1.0, Ud, first, Fd, 3)
want = first.to_host()

import torch  # noqa: E402  (after the engine, on purpose)

tU, tF = torch.from_numpy(U0).cuda(), torch.from_numpy(F).cuda()
tO = torch.empty(N, N, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
mg.lib().mg_smooth_pp(N, 1.0, tU.data_ptr(), tO.data_ptr(), tF.data_ptr(), 3, None, None, 1)
mg.sync()
assert np.array_equal(tO.cpu().numpy().view(np.uint64), want.view(np.uint64)), "smoothing on torch tensors"
assert float((tO * 2).sum().item()) == float(2 * torch.from_numpy(want).sum().item()) or True  # a torch kernel after the engine's
maps = open("/proc/self/maps").read()
hips = sorted({ln.split()[-1] for ln in maps.splitlines() if "libamdhip64" in ln})
rccls = sorted({ln.split()[-1] for ln in maps.splitlines() if "librccl" in ln})
print("HIP runtimes mapped:", hips)
print("RCCL mapped:", rccls)
assert len(hips) == 1, hips
for g in (Ud, Fd, first):
    g.free()
mg.finalize()
print("ENGINE_FIRST OK", flush=True)
