"""Subprocess worker: torch is imported FIRST (the engine library then binds to the HIP runtime
torch loaded), the engine enqueues on a torch stream and works on torch tensors."""
import os
import sys

import torch  # first, on purpose

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
os.environ.setdefault("OMP_NUM_THREADS", "4")
import numpy as np  # noqa: E402
import _oracle  # noqa: E402
import multigrid_poisson_solver_amd as mg  # noqa: E402

mg.init(0)
orc = _oracle.Oracle()
N = 256
rng = np.random.default_rng(5)
U0, F = rng.random((N, N)), rng.random((N, N)) - 0.5
st = torch.cuda.Stream()
tU, tF = torch.from_numpy(U0).cuda(), torch.from_numpy(F).cuda()
tD = torch.empty(N, N, dtype=torch.float64, device="cuda")
tO = torch.empty(N, N, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
mg.lib().mg_set_stream(st.cuda_stream)
mg.lib().mg_getResidual(N, 1.0, tU.data_ptr(), tF.data_ptr(), tD.data_ptr())
mg.lib().mg_smooth_pp(N, 1.0, tU.data_ptr(), tO.data_ptr(), tF.data_ptr(), 3, None, None, 1)
st.synchronize()
same = lambda a, b: np.array_equal(a.view(np.uint64), b.view(np.uint64))
assert same(tD.cpu().numpy(), orc.getResidual(N, 1.0, U0, F)), "residual on torch tensors"
assert same(tO.cpu().numpy(), orc.doSmoothing(N, 1.0, U0, F, 3)[0]), "smoothing on torch tensors"
mg.lib().mg_set_stream(None)
mg.finalize()
print("TORCH_INTEROP OK")
