"""Timing of unique+hull on long slices (experiments)."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_amd import _ffi as F
from pointcloud_amd.synth import make_slices
dev = torch.device("cuda", 0)
L = F.lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
p = lambda a: C.c_void_p(a.data_ptr()) if a is not None else None
E = lambda *shape, dt=torch.float64: torch.empty(*shape, dtype=dt, device=dev)
CASES = ((200, 64), (200, 512), (200, 1250), (200, 2500), (200, 5000), (200, 8192), (57, 20000), (57, 100000), (20, 1000000))
if len(sys.argv) > 2:
    CASES = ((int(sys.argv[1]), int(sys.argv[2])),)
for S, n in CASES:
    xh, yh, offh = make_slices(S, n, seed=5, outlier_frac=0.02)
    dx, dy, doff = t(xh), t(yh), t(offh); tot = len(xh)
    ux, uy, nu, hi_, nh, stt = E(tot), E(tot), E(S, dt=torch.int64), E(tot, dt=torch.int64), E(S, dt=torch.int64), E(S, dt=torch.int32)
    wsb = L.pcfit_unique_hull_workspace_bytes(tot, S); ws = E(wsb // 8 + 1, dt=torch.int64)
    fn = lambda: F.check(L.pcfit_unique_hull_batch_dev(p(dx), p(dy), p(doff), S, n, p(ux), p(uy), p(nu), p(hi_), p(nh), p(stt), None, p(ws), wsb, st))
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"unique+hull {S} x {n}: {dt*1e3:.3f} ms  ({tot/dt/1e6:.1f} M pts/s)  hull sizes {nh[:3].tolist()} kept/unique min {int(nu.min())} max {int(nu.max())}", flush=True)
