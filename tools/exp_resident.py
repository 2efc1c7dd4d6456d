"""Ten scan-level calls on a resident 1M-point scan (for a kernel trace; experiments)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloud_amd import _ffi as F, DeviceScan
from pointcloud_amd.synth import make_cylinder
NP, NS = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1_000_000, 200)
pts = make_cylinder(NP, NS, seed=0)
zc = np.arange(1.0, 2.0 * NS - 1.0 + 1e-9, 2.0)
with DeviceScan(pts) as sc:
    F.scan_hybrid_batch(sc, zc - 1, zc + 1)
    t = time.perf_counter()
    for rep in range(10):
        F.scan_hybrid_batch(sc, zc - 1, zc + 1)
    print(f"scan_hybrid_batch (resident, {NP} points, {NS} windows): {1e2*(time.perf_counter()-t):.3f} ms per call")
