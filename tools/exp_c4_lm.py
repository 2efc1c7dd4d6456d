"""Kasa + LM on 2000 resident slices of 50 000 points (config 4's shape), three calls (for counters; experiments)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_amd import _ffi as F
from pointcloud_amd.synth import make_slices
S, n = 2000, 50_000
xh, yh, offh = make_slices(S, n, seed=3, outlier_frac=float(os.environ.get("C4OUT", 0.1)))
dev = torch.device("cuda", 0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
p = lambda a: C.c_void_p(a.data_ptr())
x, y, off = t(xh), t(yh), t(offh)
L = F.lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
kasa = torch.empty(S, 3, dtype=torch.float64, device=dev); fit = torch.empty_like(kasa)
iters = torch.empty(S, 4, dtype=torch.int32, device=dev); status = torch.empty(S, dtype=torch.int32, device=dev)
for _ in range(3):
    F.check(L.pcfit_fit_batch_dev(p(x), p(y), p(off), S, n, 1.0, 100, 1e-2, 1e-10, 1e-12, p(kasa), p(fit), p(iters), p(status), st))
torch.cuda.synchronize()
print(iters[:2].tolist())
