"""Clocks per phase of unique_hull_kernel (needs the phase build: make -C pointcloud_amd/csrc phase)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_amd import _ffi as F
F.SO_PATH = os.path.join(os.path.dirname(F.SO_PATH), "libpcfit_hip_phase.so")
from pointcloud_amd.synth import make_slices
dev = torch.device("cuda", 0)
L = F.lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
L.pcfit_debug_hull_phases.argtypes = [C.c_void_p, C.c_int]
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
p = lambda a: C.c_void_p(a.data_ptr()) if a is not None else None
E = lambda *shape, dt=torch.float64: torch.empty(*shape, dtype=dt, device=dev)
for S, n in ((200, 1250), (200, 5000), (200, 8192)):
    xh, yh, offh = make_slices(S, n, seed=5, outlier_frac=0.02)
    dx, dy, doff = t(xh), t(yh), t(offh); tot = len(xh)
    ux, uy, nu, hi_, nh, stt = E(tot), E(tot), E(S, dt=torch.int64), E(tot, dt=torch.int64), E(S, dt=torch.int64), E(S, dt=torch.int32)
    wsb = L.pcfit_unique_hull_workspace_bytes(tot, S); ws = E(wsb // 8 + 1, dt=torch.int64)
    fn = lambda: F.check(L.pcfit_unique_hull_batch_dev(p(dx), p(dy), p(doff), S, n, p(ux), p(uy), p(nu), p(hi_), p(nh), p(stt), None, p(ws), wsb, st))
    fn(); out = np.zeros(8, np.int64); L.pcfit_debug_hull_phases(out.ctypes.data, 1)
    fn(); L.pcfit_debug_hull_phases(out.ctypes.data, 1)
    names = ["load", "sort", "unique", "rank", "chains"]
    print(f"{S} x {n}: " + "  ".join(f"{k} {out[i]/S/100:.0f}us" for i, k in enumerate(names)), "(clock64 at 100 MHz)")
