"""The bench's binning case on its own: a 100 M-point scan in random z order into 2000 windows through
pcfit_bin2_count_dev + pcfit_bin2_scatter_dev (two levels), timed by events; under rocprofv3 --pmc it gives the
HBM bytes of each of the binning kernels (tools/pmc_bin100m.sh).  python tools/exp_bin100m.py [points] [windows]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_amd import _ffi as F

n, S = (int(a) for a in (sys.argv[1:3] + ["100000000", "2000"])[:2])
reps = int(os.environ.get("REPS", "3"))
dev = torch.device("cuda", 0)
L = F.lib()
ptr = lambda t: C.c_void_p(t.data_ptr())
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device=dev)
g.manual_seed(1)
xyz = torch.rand(n, 3, generator=g, device=dev, dtype=torch.float64)
xyz[:, 2] *= 2.0 * S
lo = 2.0 * torch.arange(S, device=dev, dtype=torch.float64)
hi = lo + 2.0
geom = np.array([0.0, 2.0, 2.0 * (S - 1)])
wsb = L.pcfit_bin2_workspace_bytes(n, S, 0, n + n // 8)
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
off = torch.empty(S + 1, dtype=torch.int64, device=dev)
status = torch.empty(1, dtype=torch.int32, device=dev)
xo, yo = torch.empty(n, dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.float64, device=dev)


def count():
    F.check(L.pcfit_bin2_count_dev(ptr(xyz), n, ptr(lo), ptr(hi), S, F._p(geom), 0, n + n // 8, ptr(off), ptr(status),
                                   ptr(ws), wsb, stream))


def scatter():
    F.check(L.pcfit_bin2_scatter_dev(n, S, F._p(geom), 0, n + n // 8, ptr(off), ptr(xo), ptr(yo), None, ptr(ws), wsb, stream))


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


for r in range(reps):
    a, b = timed(count), timed(scatter)
    print(f"{n} points, {S} windows: level 1 + counts {a:.3f} ms, level 2 scatter {b:.3f} ms, together {a + b:.3f} ms; "
          f"members {int(off[-1].item())}, status {int(status.item())}", flush=True)
