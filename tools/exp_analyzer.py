"""End-to-end timing of the drop-in analyzer on a large scan (host arrays in, dicts out)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_amd import Config, CylinderAnalyzer, _ffi as F
from pointcloud_amd.synth import make_cylinder
N, S = int(os.environ.get("N", 20_000_000)), int(os.environ.get("S", 57))
pts = make_cylinder(N, S, seed=3, outlier_frac=0.02)
az = CylinderAnalyzer(Config())
az.analyze_with_progress(pts[:200000])
for rep in range(2):
    t0 = time.perf_counter(); res = az.analyze_with_progress(pts); dt = time.perf_counter() - t0
    print(f"analyze_with_progress {N/1e6:.0f}M pts / {len(res)} slices: {dt*1e3:.1f} ms -> {N/dt:.3e} pts/s", flush=True)
zc = az.calculate_z_centers(pts)
t0 = time.perf_counter(); off, x, y, _ = F.bin_slices(pts, zc - 1, zc + 1, want_perm=False); t1 = time.perf_counter()
h = F.hybrid_batch(x, y, off, want_profile=False); t2 = time.perf_counter()
print(f"   bin_slices (host API) {1e3*(t1-t0):.1f} ms, hybrid_batch (host API) {1e3*(t2-t1):.1f} ms, binned points {len(x)}")
t0 = time.perf_counter(); zc = az.calculate_z_centers(pts); t1 = time.perf_counter()
h = F.scan_hybrid_batch(pts, zc - 1, zc + 1); t2 = time.perf_counter()
print(f"   calculate_z_centers {1e3*(t1-t0):.1f} ms, scan_hybrid_batch {1e3*(t2-t1):.1f} ms")
import ctypes as C
d = torch.empty((N, 3), dtype=torch.float64, device="cuda")
for rep in range(2):
    t0 = time.perf_counter(); F.check(F.lib().pcfit_memcpy_h2d(C.c_void_p(d.data_ptr()), pts.ctypes.data, pts.nbytes, None)); F.check(F.lib().pcfit_stream_synchronize(None)); t1 = time.perf_counter()
    print(f"   hipMemcpyAsync H2D of the scan on the null stream: {1e3*(t1-t0):.1f} ms = {pts.nbytes/(t1-t0)/1e9:.1f} GB/s")
