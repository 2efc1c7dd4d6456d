"""Many thin windows through the scan-level call (robustness check; experiments)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_amd import _ffi as F, Config, CylinderAnalyzer
from pointcloud_amd.synth import make_cylinder
from oracle import ref_np as R
pts = make_cylinder(2_000_000, 200, seed=0)
CylinderAnalyzer(Config()).analyze_with_progress(pts)          # (library and device start-up)
for step in (0.08, 0.025):
    az = CylinderAnalyzer(Config(z_step=step, min_points_per_slice=50))
    az.analyze_with_progress(pts)
    t = time.perf_counter(); zc = az.calculate_z_centers(pts); h = F.scan_hybrid_batch(pts, zc - step / 2, zc + step / 2); dt = time.perf_counter() - t
    print(f"z_step {step}: library call alone {dt*1e3:.1f} ms")
    t = time.perf_counter(); res = az.analyze_with_progress(pts); dt = time.perf_counter() - t
    zc = az.calculate_z_centers(pts)
    print(f"z_step {step}: {len(zc)} windows, {len(res)} results, {dt*1e3:.1f} ms")
    k = len(res) // 2
    r = res[k]
    ref = R.hybrid_fit(pts, r["z_center"], step / 2)
    err = max(abs(r["cx"] - ref["center_x"]), abs(r["cy"] - ref["center_y"]), abs(r["R"] - ref["radius"]))
    print(f"   window {k}: n {r['n_points_slice']}, |fit - restatement| {err:.2e}")
