"""Upload of a scan from pageable host memory into a resident DeviceScan (pcfit_scan_upload), timed by the host
clock: python tools/exp_upload.py [points]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloud_amd import DeviceScan
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
pts = np.random.default_rng(0).normal(size=(n, 3))
for rep in range(4):
    t = time.perf_counter()
    with DeviceScan(pts) as sc:
        dt = time.perf_counter() - t
    print(f"upload of {pts.nbytes / 1e9:.2f} GB: {dt * 1e3:.1f} ms = {pts.nbytes / dt / 1e9:.1f} GB/s", flush=True)
