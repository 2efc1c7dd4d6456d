"""Where the time of the scan-level call goes on a small scan (experiments)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloud_amd import _ffi as F, DeviceScan
from pointcloud_amd.synth import make_cylinder
T = time.perf_counter
pts = make_cylinder(1_000_000, 200, seed=0)
zc = np.arange(1.0, 399.0 + 1e-9, 2.0)
h = F.scan_hybrid_batch(pts, zc - 1, zc + 1)
m = np.diff(h["cand_offsets"])
print("distinct candidate counts:", len(np.unique(m)), "of", len(m), "slices")
t = T()
for k in np.unique(m):
    F.choice3_table(int(k), 4000)
print(f"sampling tables on the host: {1e3*(T()-t):.2f} ms")
for rep in range(3):
    t = T(); F.scan_hybrid_batch(pts, zc - 1, zc + 1); print(f"scan_hybrid_batch (array): {1e3*(T()-t):.2f} ms")
with DeviceScan(pts) as sc:
    for rep in range(3):
        t = T(); F.scan_hybrid_batch(sc, zc - 1, zc + 1); print(f"scan_hybrid_batch (resident): {1e3*(T()-t):.2f} ms")
