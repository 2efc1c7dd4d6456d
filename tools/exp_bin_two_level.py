"""Binning through the host-level call (pcfit_scan_bin_slices): digest of offsets | x | y | perm and time, for
comparing PCFIT_BIN_TWO_LEVEL_MIN=-1 (one level) with the default (experiments)."""
import hashlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_amd import _ffi as F, DeviceScan

for N, S, width in ((30_000_000, 700, 1.0), (8_000_000, 257, 1.0), (20_000_000, 2000, 0.5), (5_000_000, 513, 1.0)):
    for mode in ("random", "sorted", "edges"):
        g = np.random.default_rng(N + S)
        pts = g.random((N, 3))
        step = 2.0
        if mode == "random":
            pts[:, 2] = pts[:, 2] * (step * S + 2.0) - 1.0
        elif mode == "sorted":
            pts[:, 2] = np.arange(N) * (step * S / N) + pts[:, 2] * 0.01
        else:
            pts[:, 2] = np.round(pts[:, 2] * (step * S))          # half the points exactly on window edges
        lo = step * np.arange(S)
        hi = lo + step * width
        with DeviceScan(pts) as sc:
            for want_perm in (False, True):
                F.bin_slices(sc, lo[:3], hi[:3], want_perm=False)
                t = time.perf_counter()
                off, x, y, perm = F.bin_slices(sc, lo, hi, want_perm=want_perm)
                dt = time.perf_counter() - t
                h = hashlib.sha1()
                for a in (off, x, y) + ((perm,) if want_perm else ()):
                    h.update(a.tobytes())
                print(f"N={N:>9} S={S:>5} w={width} {mode:7s} perm={int(want_perm)} total={off[-1]:>10} "
                      f"{1e3 * dt:8.1f} ms (with the download)  {h.hexdigest()[:12]}", flush=True)
