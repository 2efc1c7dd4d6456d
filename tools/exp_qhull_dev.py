"""Time of the device replay of Qhull's construction on slices of the bench's kind, with the number of insertions and
of point visits it took and the slices it handed back, by cause (needs a GPU).
python tools/exp_qhull_dev.py [slices] [points per slice] [x0 y0 (where the cylinder stands)] [decimals the coordinates
are rounded to, or - for none] [fraction of outliers]"""
import ctypes as C
import sys
import time

import numpy as np
import torch

import os

sys.path.insert(0, ".")
from pointcloud_amd import _ffi as F

PHASES = os.environ.get("PHASES") == "1"            # the phase-clock build (make -C pointcloud_amd/csrc phase)
if PHASES:
    F.SO_PATH = F.SO_PATH.replace("libpcfit_hip.so", "libpcfit_hip_phase.so")

S, n = (int(a) for a in (sys.argv[1:3] + ["2000", "50000"])[:2])
shift = (float(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (3.0, -2.0)      # where the cylinder stands
g = np.random.default_rng(0)
dev = torch.device("cuda", 0)
th = torch.rand(S * n, dtype=torch.float64, device=dev) * (2 * np.pi)
r = 10 + 0.05 * torch.randn(S * n, dtype=torch.float64, device=dev)
bad = torch.rand(S * n, device=dev) < (float(sys.argv[6]) if len(sys.argv) > 6 else 0.1)
r = torch.where(bad, torch.rand(S * n, dtype=torch.float64, device=dev) * 14, r)
x, y = shift[0] + r * torch.cos(th), shift[1] + r * torch.sin(th)
decimals = int(sys.argv[5]) if len(sys.argv) > 5 and sys.argv[5] != "-" else None          # coordinates as a text file with that many decimals holds them
if decimals is not None:
    x, y = torch.round(x * 10 ** decimals) / 10 ** decimals, torch.round(y * 10 ** decimals) / 10 ** decimals
off = torch.arange(S + 1, dtype=torch.int64, device=dev) * n
p = lambda a: C.c_void_p(a.data_ptr())
# hull sizes from the library (sorted route, once)
h = F.unique_hull_batch(x.cpu().numpy(), y.cpu().numpy(), off.cpu().numpy())
nh = torch.from_numpy(h["n_hull"]).to(dev)
# the hulls themselves (HULL=1): the kernel takes the coordinate extremes from them and picks its form by the largest
HULL = os.environ.get("HULL", "1") == "1"
hx, hy, hi = (torch.from_numpy(h[k]).to(dev) for k in ("ux", "uy", "hull_idx"))
hull_args = (p(hx), p(hy), p(hi), int(h["n_hull"].max())) if HULL else (None, None, None, -1)
scratch = torch.empty(S * n * 48, dtype=torch.uint8, device=dev)
start = torch.empty(S, dtype=torch.int64, device=dev)
why = np.zeros(16, np.int64)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for rep in range(3):
    F.check(F.lib().pcfit_debug_qhull_reasons(F._p(why), 1))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    F.check(F.lib().pcfit_qhull_start_batch_dev(p(x), p(y), p(off), None, p(nh), None, S, S * n, p(scratch), S * n * 48,
                                                p(start), *hull_args, st))
    e1.record()
    torch.cuda.synchronize()
    F.check(F.lib().pcfit_debug_qhull_reasons(F._p(why), 1))
    print(f"{e0.elapsed_time(e1):.3f} ms; per slice: {why[14] / S:.1f} insertions, {why[15] / S / n:.2f} visits per point after the "
          f"first partition; handed back {int(why[:14].sum())} {why[:14].tolist()}; mean hull {float(nh.double().mean()):.1f}, "
          f"largest {int(nh.max())}")
if PHASES:
    out = np.zeros(8, np.int64)
    L = F.lib()
    L.pcfit_debug_qhull_phase_clocks.argtypes = [C.c_void_p, C.c_int]
    L.pcfit_debug_qhull_phase_clocks(F._p(out), 1)
    F.check(L.pcfit_qhull_start_batch_dev(p(x), p(y), p(off), None, p(nh), None, S, S * n, p(scratch), S * n * 48, p(start), *hull_args, st))
    L.pcfit_debug_qhull_phase_clocks(F._p(out), 0)
    sampled = (S + 15) // 16                       # one workgroup in 16 reports; wall_clock64 ticks at 100 MHz
    names = ["extremes", "simplex", "first partition", "placing", "the list's thread", "partition of visible sets",
             "farthest points", "end"]
    for k, nm in enumerate(names):
        print(f"   phase {nm:26s} {out[k] / sampled / 100:8.2f} us/slice  {100 * out[k] / max(out.sum(), 1):5.1f} %")
