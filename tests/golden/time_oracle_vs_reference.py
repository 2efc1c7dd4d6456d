#!/usr/bin/env python3
"""Time the numpy restatement next to the real reference on the same scan (build container only; the GPU box
has no reference).  Run:  MPLBACKEND=Agg python tests/golden/time_oracle_vs_reference.py [points] [windows]

Both run their per-window loop serially on one thread: reference CylinderAnalyzer.process_slice
(analyzer.py:46-129) against oracle.ref_np.process_slice, window by window, and the results are compared."""
import os
import sys
import tempfile
import time
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
warnings.simplefilter("ignore")
os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
sys.path.insert(0, "/root/reference")
sys.path.insert(1, ROOT)
os.chdir(tempfile.mkdtemp(prefix="pcfit_time_"))        # the analyzer creates ./plots_slices

import analyzer as ran                                   # noqa: E402  (reference)
from oracle import ref_np as R                           # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 40
pts = R.make_cylinder(N, S, seed=0)
az = ran.CylinderAnalyzer(ran.Config())
zc = az.calculate_z_centers(pts)
t0 = time.perf_counter()
ref = [az.process_slice(pts, z) for z in zc]
t1 = time.perf_counter()
mine = [R.process_slice(pts, z) for z in zc]
t2 = time.perf_counter()
worst = 0.0
for (r, _, wr), (m, _, wm) in zip(ref, mine):
    assert (wr is None) == (wm is None)
    if wr is None:
        res = r[0]
        for key in ("cx", "cy", "R", "Rmax", "Rmin"):
            worst = max(worst, abs(res[key] - m[key]) / max(1.0, abs(res[key])))
print(f"{N} points, {len(zc)} windows: reference {t1 - t0:.2f} s, restatement {t2 - t1:.2f} s "
      f"(ratio {(t2 - t1) / (t1 - t0):.2f}); largest relative difference of cx, cy, R, Rmax, Rmin {worst:.1e}")
