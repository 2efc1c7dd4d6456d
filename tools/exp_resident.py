"""Ten scan-level calls on a resident scan (for a kernel trace; experiments).
python tools/exp_resident.py [points] [windows] [fraction of outliers] [order: qhull | canonical] [decimals the
coordinates are rounded to, as a text file holds them]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloud_amd import _ffi as F, DeviceScan
from pointcloud_amd.synth import make_cylinder
NP, NS = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1_000_000, 200)
FRAC = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
ORDER = sys.argv[4] if len(sys.argv) > 4 else "qhull"
pts = make_cylinder(NP, NS, seed=0, outlier_frac=FRAC)
if len(sys.argv) > 5:
    pts = np.round(pts, int(sys.argv[5]))
zc = np.arange(1.0, 2.0 * NS - 1.0 + 1e-9, 2.0)
with DeviceScan(pts) as sc:
    why = np.zeros(16, np.int64)
    F.check(F.lib().pcfit_debug_qhull_reasons(F._p(why), 1))
    F.scan_hybrid_batch(sc, zc - 1, zc + 1, order=ORDER)
    F.check(F.lib().pcfit_debug_qhull_reasons(F._p(why), 1))
    print("slices the device replay handed back, by cause:", why[:14].tolist())
    t = time.perf_counter()
    REPS = int(os.environ.get("REPS", "10"))
    for rep in range(REPS):
        F.scan_hybrid_batch(sc, zc - 1, zc + 1, order=ORDER)
    print(f"scan_hybrid_batch (resident, {NP} points, {NS} windows, outliers {FRAC}, {ORDER} order): {1e3*(time.perf_counter()-t)/REPS:.3f} ms per call")
