"""Kernel timing survey on the configurations of BASELINE.json (experiments; not the bench contract)."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_amd import _ffi as F
from pointcloud_amd.synth import make_slices, make_cylinder

dev = torch.device("cuda", 0)
L = F.lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
p = lambda a: C.c_void_p(a.data_ptr()) if a is not None else None
E = lambda *shape, dt=torch.float64: torch.empty(*shape, dtype=dt, device=dev)

def timeit(name, fn, reps=5, units=None):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    extra = "" if units is None else f"  {units[0]/ms/1e6:10.2f} {units[1]}"
    print(f"{name:44s} {ms:9.3f} ms{extra}", flush=True)
    return ms

which = sys.argv[1:] or ["c3", "c4", "c1"]

if "c3" in which:
    S, n, T = 10000, 8192, 1024
    xh, yh, offh = make_slices(S, n, seed=2, outlier_frac=0.3)
    x, y, off = t(xh), t(yh), t(offh)
    state = F.seed_state(42); tab = F.choice3_table(n, T)
    dtab, dof = t(tab.ravel()), torch.zeros(S, dtype=torch.int32, device=dev)
    model, score, best, mask, status = E(S, 3), E(S, dt=torch.int32), E(S, dt=torch.int32), E(S * n, dt=torch.uint8), E(S, dt=torch.int32)
    ws = E(S, dt=torch.int64)
    pts_hyp = S * n * T
    timeit("C3 ransac 10k x 8192 x 1024 hyp", lambda: F.check(L.pcfit_ransac_batch_dev(p(x), p(y), p(off), S, 0.5, T, 0.1, None, p(dtab), p(dof), p(model), p(score), p(best), p(mask), p(status), p(ws), S * 8, st)), reps=3, units=(pts_hyp / 1e3, "T point-hyp/s"))
    del x, y, mask

if "c4" in which:
    N, S = int(os.environ.get("C4N", 100_000_000)), int(os.environ.get("C4S", 2000))
    pts = make_cylinder(N, S, seed=3, outlier_frac=float(os.environ.get("C4OUT", 0.1)))
    xyz = t(pts); del pts
    zc = np.arange(1.0, 2.0 * S - 1.0 + 1e-9, 2.0); lo, hi = zc - 1.0, zc + 1.0
    dlo, dhi = t(lo), t(hi)
    geom = np.array([lo[0], hi[0], lo[-1]]); 
    wsb = L.pcfit_bin_workspace_bytes(N, S); ws = E(wsb // 8 + 1, dt=torch.int64)
    off = E(S + 1, dt=torch.int64)
    timeit(f"C4 bin count ({N//1000000}M pts, {S} windows)", lambda: F.check(L.pcfit_bin_count_dev(p(xyz), N, p(dlo), p(dhi), S, geom.ctypes.data, p(off), p(ws), wsb, st)), units=(N * 24 / 1e3, "GB/s"))
    tot = int(off[-1].item()); xo, yo = E(tot), E(tot)
    timeit("C4 bin scatter", lambda: F.check(L.pcfit_bin_scatter_dev(p(xyz), N, p(dlo), p(dhi), S, geom.ctypes.data, p(off), p(xo), p(yo), None, p(ws), wsb, st)), units=(N * 40 / 1e3, "GB/s"))
    maxlen = int((off[1:] - off[:-1]).max().item())
    kasa, fit, iters, status = E(S, 3), E(S, 3), E(S, 4, dt=torch.int32), E(S, dt=torch.int32)
    timeit("C4 kasa (50k pts/slice, streamed)", lambda: F.check(L.pcfit_kasa_batch_dev(p(xo), p(yo), p(off), S, maxlen, p(kasa), p(status), st)), units=(tot * 16 / 1e3, "GB/s"))
    timeit("C4 kasa+lm (streamed from L2/HBM)", lambda: F.check(L.pcfit_fit_batch_dev(p(xo), p(yo), p(off), S, maxlen, 1.0, 100, 1e-2, 1e-10, 1e-12, p(kasa), p(fit), p(iters), p(status), st)), units=(tot / 1e3, "M pts/s"))
    print("   sweeps", iters[:3].tolist())
    sel = E(S, 720, dt=torch.int64)
    timeit("C4 angle_max 720 bins", lambda: F.check(L.pcfit_angle_max_batch_dev(p(xo), p(yo), p(off), S, p(fit), 720, p(sel), st)), units=(tot * 16 / 1e3, "GB/s"))
    H = 256; edges = t(np.linspace(-0.5, 0.5, H + 1)); hist, rmm, amm = E(S, H, dt=torch.int64), E(S, 2), E(S, 2, dt=torch.int64)
    timeit("C4 residual hist", lambda: F.check(L.pcfit_residuals_batch_dev(p(xo), p(yo), p(off), S, p(fit), None, None, None, H, p(edges), p(hist), p(rmm), p(amm), st)), units=(tot * 16 / 1e3, "GB/s"))
    swb = L.pcfit_split_workspace_bytes(tot, S, 720); sws = E(swb // 8 + 1, dt=torch.int64)
    timeit("C4 split kasa", lambda: F.check(L.pcfit_kasa_split_dev(p(xo), p(yo), p(off), S, tot, p(kasa), p(status), p(sws), swb, st)), units=(tot * 32 / 1e3, "GB/s"))
    timeit("C4 split kasa+lm", lambda: F.check(L.pcfit_fit_split_dev(p(xo), p(yo), p(off), S, tot, 1.0, 100, 1e-2, 1e-10, 1e-12, p(kasa), p(fit), p(iters), p(status), p(sws), swb, st)), units=(tot / 1e3, "M pts/s"))
    print("   sweeps", iters[:3].tolist())
    timeit("C4 split angle_max 720 bins", lambda: F.check(L.pcfit_angle_max_split_dev(p(xo), p(yo), p(off), S, tot, p(fit), 720, p(sel), p(sws), swb, st)), units=(tot * 32 / 1e3, "GB/s"))
    timeit("C4 split residual hist", lambda: F.check(L.pcfit_residuals_split_dev(p(xo), p(yo), p(off), S, tot, p(fit), None, None, None, H, p(edges), p(hist), p(rmm), p(amm), p(sws), swb, st)), units=(tot * 16 / 1e3, "GB/s"))
    del xyz, xo, yo

if "c1" in which:
    pts = make_cylinder(1_000_000, 200, seed=0)
    from pointcloud_amd import Config, CylinderAnalyzer
    az = CylinderAnalyzer(Config())
    az.analyze_with_progress(pts)
    t0 = time.perf_counter(); res = az.analyze_with_progress(pts); dt = time.perf_counter() - t0
    print(f"C1 analyze_with_progress 1M pts / 200 windows (host arrays in, dicts out): {dt*1e3:.1f} ms -> {1e6/dt:.3e} pts/s, {len(res)} results")
    zc = az.calculate_z_centers(pts)
    t0 = time.perf_counter(); off, x, y, _ = F.bin_slices(pts, zc - 1, zc + 1, want_perm=False); t1 = time.perf_counter()
    h = F.hybrid_batch(x, y, off, want_profile=False); t2 = time.perf_counter()
    print(f"   bin_slices (host API) {1e3*(t1-t0):.1f} ms, hybrid_batch (host API) {1e3*(t2-t1):.1f} ms")
    # device-side pieces of the hybrid
    dx, dy, doff = t(x), t(y), t(off); S = len(off) - 1; tot = len(x); maxlen = int(np.diff(off).max())
    ux, uy, nu, hi_, nh, stt = E(tot), E(tot), E(S, dt=torch.int64), E(tot, dt=torch.int64), E(S, dt=torch.int64), E(S, dt=torch.int32)
    wsb = L.pcfit_unique_hull_workspace_bytes(tot, S); ws = E(wsb // 8 + 1, dt=torch.int64)
    timeit("C1 unique+hull 200 x 5000", lambda: F.check(L.pcfit_unique_hull_batch_dev(p(dx), p(dy), p(doff), S, maxlen, p(ux), p(uy), p(nu), p(hi_), p(nh), p(stt), None, p(ws), wsb, st)))
    print("   hull sizes", nh[:5].tolist())
    xh, yh, offh = make_slices(2000, 8192, seed=5); dx, dy, doff = t(xh), t(yh), t(offh); S = 2000; tot = len(xh)
    ux, uy, nu, hi_, nh, stt = E(tot), E(tot), E(S, dt=torch.int64), E(tot, dt=torch.int64), E(S, dt=torch.int64), E(S, dt=torch.int32)
    wsb = L.pcfit_unique_hull_workspace_bytes(tot, S); ws = E(wsb // 8 + 1, dt=torch.int64)
    timeit("unique+hull 2000 x 8192", lambda: F.check(L.pcfit_unique_hull_batch_dev(p(dx), p(dy), p(doff), S, 8192, p(ux), p(uy), p(nu), p(hi_), p(nh), p(stt), None, p(ws), wsb, st)), reps=2, units=(tot / 1e3, "M pts/s"))
