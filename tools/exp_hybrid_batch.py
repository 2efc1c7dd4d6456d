"""The hybrid chain on a large batch of medium slices (experiments)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_amd import _ffi as F
from pointcloud_amd.synth import make_slices
S, n = int(os.environ.get("S", 2000)), int(os.environ.get("NPTS", 8192))
x, y, off = make_slices(S, n, seed=5, outlier_frac=0.05)
for unique in (False, True):
    F.hybrid_batch(x, y, off, want_profile=False, want_unique=unique)
    t = time.perf_counter(); h = F.hybrid_batch(x, y, off, want_profile=False, want_unique=unique); dt = time.perf_counter() - t
    print(f"hybrid_batch {S} x {n} (host arrays, distinct points {'returned' if unique else 'not returned'}): "
          f"{dt*1e3:.1f} ms, ok {int((h['status'] == 0).sum())}, candidates {h['cand_offsets'][-1]}")
