"""The angular-bin boundary and the residual statistics on 2000 pre-binned slices of ~50 000 points (config 4's windows),
separately and in one sweep: event timings.  python tools/exp_boundary.py [slices] [points] [reps]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_amd import _ffi as F

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(3)
N = S * n
ang = torch.rand(N, generator=g, device=dev, dtype=torch.float64) * (2 * np.pi)
r = 10.0 + 0.05 * torch.randn(N, generator=g, device=dev, dtype=torch.float64)
bad = torch.rand(N, generator=g, device=dev, dtype=torch.float64) < 0.1
r = torch.where(bad, torch.rand(N, generator=g, device=dev, dtype=torch.float64) * 14.0, r)
x, y = 3.0 + r * torch.cos(ang), -2.0 + r * torch.sin(ang)
del ang, r, bad
off = torch.arange(S + 1, device=dev, dtype=torch.int64) * n
fit = torch.tensor([[3.001, -2.002, 10.0]], dtype=torch.float64, device=dev).repeat(S, 1).contiguous()
H, bins = 256, 720
edges = torch.linspace(-0.5, 0.5, H + 1, dtype=torch.float64, device=dev)
sel, sel1 = (torch.empty(S, bins, dtype=torch.int64, device=dev) for _ in range(2))
hist, hist1 = (torch.empty(S, H, dtype=torch.int64, device=dev) for _ in range(2))
rmm = torch.empty(S, 2, dtype=torch.float64, device=dev)
amm, amm1 = (torch.empty(S, 2, dtype=torch.int64, device=dev) for _ in range(2))
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
L = F.lib()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timed(fn):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


a = timed(lambda: F.check(L.pcfit_angle_max_batch_dev(p(x), p(y), p(off), S, p(fit), bins, p(sel), st)))
b = timed(lambda: F.check(L.pcfit_residuals_batch_dev(p(x), p(y), p(off), S, p(fit), None, None, None, H, p(edges), p(hist),
                                                      p(rmm), p(amm), st)))
c = timed(lambda: F.check(L.pcfit_boundary_residuals_batch_dev(p(x), p(y), p(off), S, p(fit), bins, p(sel1), H, p(edges),
                                                               p(hist1), p(rmm), p(amm1), st)))
same = torch.equal(sel, sel1) and torch.equal(hist, hist1) and torch.equal(amm, amm1)
gb = 16.0 * N / 1e9
print(f"{S} x {n}: boundary {a:.3f} ms ({gb / a:.2f} TB/s)  statistics {b:.3f} ms ({gb / b:.2f} TB/s)  "
      f"one sweep {c:.3f} ms ({gb / c:.2f} TB/s)  equal {same}")
