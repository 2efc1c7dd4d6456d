"""Several scans one after the other on one GPU: plain loop against a loop that uploads the next scan from a
second thread while the current one is analysed (experiments; BASELINE's multi-scan configuration in small)."""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_amd import Config, CylinderAnalyzer, DeviceScan
from pointcloud_amd.synth import make_cylinder
N, S, K = int(os.environ.get("N", 20_000_000)), int(os.environ.get("S", 100)), int(os.environ.get("K", 6))
scans = [make_cylinder(N, S, seed=100 + k, outlier_frac=0.02) for k in range(K)]
az = CylinderAnalyzer(Config())
az.analyze_with_progress(scans[0])
t0 = time.perf_counter()
a = [az.analyze_with_progress(p) for p in scans]
t1 = time.perf_counter()
with ThreadPoolExecutor(1) as pool:
    nxt = pool.submit(DeviceScan, scans[0])
    b = []
    for k in range(K):
        cur = nxt.result()
        if k + 1 < K:
            nxt = pool.submit(DeviceScan, scans[k + 1])
        b.append(az.analyze_with_progress(cur))
        cur.close()
t2 = time.perf_counter()
assert all(x == y for x, y in zip(a, b))
print(f"{K} scans x {N/1e6:.0f}M points, {S} windows: loop {1e3*(t1-t0)/K:.1f} ms per scan "
      f"({N*K/(t1-t0):.3e} pts/s), upload of the next scan overlapped {1e3*(t2-t1)/K:.1f} ms per scan "
      f"({N*K/(t2-t1):.3e} pts/s)")
