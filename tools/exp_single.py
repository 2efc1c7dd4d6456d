"""Latency of the single-slice entry points (experiments)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_amd import Config, CylinderAnalyzer, DeviceScan
from pointcloud_amd.utils.circle_fitting import fit_circle_hybrid_at_z, fit_circle_pratt, lm_circle_geometric
from pointcloud_amd.synth import make_cylinder
pts = make_cylinder(1_000_000, 200, seed=0)
sl = pts[(pts[:, 2] >= 100) & (pts[:, 2] <= 102)]
def timeit(name, fn, reps=20):
    fn()
    t = time.perf_counter()
    for _ in range(reps): fn()
    print(f"{name}: {1e3*(time.perf_counter()-t)/reps:.3f} ms")
timeit(f"fit_circle_hybrid_at_z on a {len(sl)}-point slice", lambda: fit_circle_hybrid_at_z(sl, 101.0, 1.0))
timeit("fit_circle_pratt", lambda: fit_circle_pratt(sl[:, 0], sl[:, 1]))
timeit("lm_circle_geometric", lambda: lm_circle_geometric(sl[:, 0], sl[:, 1], 3.0, -2.0, 10.0))
az = CylinderAnalyzer(Config())
timeit("process_slice on the 1M-point array", lambda: az.process_slice(pts, 101.0), reps=10)
with DeviceScan(pts) as scan:
    timeit("process_slice on the resident scan", lambda: az.process_slice(scan, 101.0), reps=10)
