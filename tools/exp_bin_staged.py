"""Binning scatter: time and a digest of the outputs, for comparing PCFIT_BIN_STAGED=0 / 1 (experiments)."""
import ctypes as C, hashlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_amd import _ffi as F

dev = torch.device("cuda", 0)
L = F.lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda a: C.c_void_p(a.data_ptr()) if a is not None else None
E = lambda *shape, dt=torch.float64: torch.empty(*shape, dtype=dt, device=dev)


def run(N, S, mode, want_perm):
    g = torch.Generator(device=dev); g.manual_seed(N + S)
    xyz = torch.rand(N, 3, dtype=torch.float64, device=dev, generator=g)
    if mode == "random":
        xyz[:, 2] = xyz[:, 2] * (2.0 * S + 2.0) - 1.0
    elif mode == "sorted":                        # a scanner that climbs: z follows the point index
        xyz[:, 2] = torch.arange(N, dtype=torch.float64, device=dev) * (2.0 * S / N) + xyz[:, 2] * 0.01
    elif mode == "edges":                         # half the points exactly on window edges
        xyz[:, 2] = torch.round(xyz[:, 2] * (2.0 * S))
    zc = np.arange(1.0, 2.0 * S - 1.0 + 1e-9, 2.0); lo, hi = zc - 1.0, zc + 1.0
    dlo, dhi = torch.from_numpy(lo).to(dev), torch.from_numpy(hi).to(dev)
    geom = np.array([lo[0], hi[0], lo[-1]])
    wsb = L.pcfit_bin_workspace_bytes(N, S); ws = E(wsb // 8 + 1, dt=torch.int64)
    off = E(S + 1, dt=torch.int64)
    F.check(L.pcfit_bin_count_dev(p(xyz), N, p(dlo), p(dhi), S, geom.ctypes.data, p(off), p(ws), wsb, st))
    tot = int(off[-1].item())
    xo, yo = E(tot), E(tot)
    perm = E(tot, dt=torch.int64) if want_perm else None
    xo.fill_(-7.0); yo.fill_(-7.0)
    fn = lambda: F.check(L.pcfit_bin_scatter_dev(p(xyz), N, p(dlo), p(dhi), S, geom.ctypes.data, p(off), p(xo), p(yo),
                                                 p(perm), p(ws), wsb, st))
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    h = hashlib.sha1()
    for a in (off, xo, yo) + ((perm,) if want_perm else ()):
        h.update(a.cpu().numpy().tobytes())
    print(f"N={N:>10} S={S:>4} {mode:7s} perm={int(want_perm)} total={tot:>10} scatter {ms:8.3f} ms "
          f"{(N * 24 + tot * (24 if want_perm else 16)) / ms / 1e6:8.1f} GB/s  {h.hexdigest()[:12]}", flush=True)


CASES = ((50_000_000, 57), (20_000_000, 200), (1_000_000, 57), (100_003, 7), (5_000_000, 256))
if len(sys.argv) > 2:
    CASES = ((int(sys.argv[1]), int(sys.argv[2])),)
for N, S in CASES:
    for mode in ("random", "sorted", "edges"):
        for want_perm in (False, True):
            run(N, S, mode, want_perm)
