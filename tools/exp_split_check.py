"""Split LM against the one-workgroup LM on long slices: results, sweep counts, time (experiments)."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_amd import _ffi as F
from pointcloud_amd.synth import make_cylinder
dev = torch.device("cuda", 0)
L = F.lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
p = lambda a: C.c_void_p(a.data_ptr()) if a is not None else None
E = lambda *shape, dt=torch.float64: torch.empty(*shape, dtype=dt, device=dev)
N, S = int(os.environ.get("C4N", 20_000_000)), int(os.environ.get("C4S", 57))
pts = make_cylinder(N, S, seed=3, outlier_frac=0.1)
zc = np.arange(1.0, 2.0 * S - 1.0 + 1e-9, 2.0)
off, x, y, _ = F.bin_slices(pts, zc - 1, zc + 1, want_perm=False)
xo, yo, doff = t(x), t(y), t(off); tot = len(x); maxlen = int(np.diff(off).max())
kasa, fit, iters, status = E(S, 3), E(S, 3), E(S, 4, dt=torch.int32), E(S, dt=torch.int32)
fit2, iters2 = E(S, 3), E(S, 4, dt=torch.int32)
F.check(L.pcfit_fit_batch_dev(p(xo), p(yo), p(doff), S, maxlen, 1.0, 100, 1e-2, 1e-10, 1e-12, p(kasa), p(fit), p(iters), p(status), st))
swb = L.pcfit_split_workspace_bytes(tot, S, 0); sws = E(swb // 8 + 1, dt=torch.int64)
fit2.fill_(float("nan")); iters2.fill_(-7)
fn = lambda: F.check(L.pcfit_fit_split_dev(p(xo), p(yo), p(doff), S, tot, 1.0, 100, 1e-2, 1e-10, 1e-12, p(kasa), p(fit2), p(iters2), p(status), p(sws), swb, st))
fn(); torch.cuda.synchronize()
t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
d = (fit2 - fit).abs().max().item()
print(f"split fit: {dt*1e3:.2f} ms, max |difference| to the one-workgroup fit {d:.3e}")
print("sweeps per slice (split):", (iters2[:, 2] + iters2[:, 3]).tolist())
print("unfinished:", int((iters2[:, 0] == -7).sum().item()), "NaN results:", int(torch.isnan(fit2).any(dim=1).sum().item()))
