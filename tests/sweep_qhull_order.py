"""Randomised comparison of the library's host-side replay of Qhull's 2-D construction (pcfit_qhull_order_batch)
with scipy.spatial.ConvexHull(xy).vertices (build container only: needs scipy; no GPU work).  Not collected by
pytest:  python tests/sweep_qhull_order.py [slices] [seed]"""
import os
import sys
import time

import numpy as np
from scipy.spatial import ConvexHull

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.qhull2d import KINDS, make_case     # noqa: E402  (inputs only)
from pointcloud_amd import _ffi as F            # noqa: E402


def main(n_slices=6000, seed=0):
    gen = np.random.default_rng(seed)
    xs, ys, off, want, kinds = [], [], [0], [], []
    for r in range(n_slices):
        kind = KINDS[r % len(KINDS)]
        xy = make_case(kind, int(gen.choice([3, 4, 6, 10, 30, 200, 1500, 8192, 30000])), int(gen.integers(1 << 30)))
        if len(xy) < 3:
            continue
        try:
            v = ConvexHull(xy).vertices.astype(np.int64)
        except Exception:           # noqa: BLE001  (Qhull refuses flat input)
            continue
        xs.append(xy[:, 0]), ys.append(xy[:, 1]), off.append(off[-1] + len(xy)), want.append(v), kinds.append(kind)
    x, y, off = np.concatenate(xs), np.concatenate(ys), np.array(off, np.int64)
    t0 = time.perf_counter()
    r = F.qhull_order_batch(x, y, off)
    dt = time.perf_counter() - t0
    tally = {k: [0, 0, 0, 0] for k in KINDS}
    for s, v in enumerate(want):
        same = np.array_equal(r["walk"][off[s]:off[s] + r["n_walk"][s]], v)
        t = tally[kinds[s]]
        t[0] += 1
        t[1] += bool(r["sure"][s]) and not same
        t[2] += not r["sure"][s]
        t[3] += (not r["sure"][s]) and not same
    print(f"{len(want)} slices, {x.size} points, replay {dt:.3f} s ({x.size / dt / 1e6:.1f} M points/s)")
    print("kind: slices, wrong while sure, not sure, wrong while not sure")
    for k, t in tally.items():
        print(f"  {k:8s} {t}")
    tot = np.sum(list(tally.values()), axis=0)
    print("  total   ", tot.tolist())
    return int(tot[1])


if __name__ == "__main__":
    sys.exit(1 if main(*(int(a) for a in sys.argv[1:3])) else 0)
