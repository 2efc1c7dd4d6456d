"""Makes tests/golden/qhull_far_vertex.npz (build container: needs scipy): the slice that tests/sweep_qhull_device.py
draws as number 170 of round 1 from seed 31 - replayed here draw for draw - as its sorted distinct points, with
scipy.spatial.ConvexHull(xy).vertices.  python tests/golden/gen_qhull_far_vertex.py"""
import os
import sys

import numpy as np
from scipy.spatial import ConvexHull

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.qhull2d import KINDS, make_case     # noqa: E402


def slice_of(seed, want_round, want_slice):
    gen = np.random.default_rng(seed)
    for r in range(want_round + 1):
        for k in range(int(gen.integers(50, 600))):
            n = int(gen.choice([3, 4, 6, 10, 30, 200, 1500, 8192, 9000, 30000, 70000]))
            if gen.random() < 0.25:
                th = gen.uniform(0, 2 * np.pi, n)
                rad = 10 + gen.normal(0, 0.005, n)
                xy = np.unique(np.round(np.stack([3 + rad * np.cos(th), -2 + rad * np.sin(th)], 1), int(gen.choice([2, 3, 3, 4]))),
                               axis=0)
            else:
                xy = make_case(KINDS[int(gen.integers(len(KINDS)))], n, int(gen.integers(1 << 30)))
            if gen.random() < 0.2:
                xy = np.concatenate([xy, xy[gen.integers(0, len(xy), len(xy) // 3 + 1)]])
            elif gen.random() < 0.15:
                xy = np.repeat(xy, int(gen.integers(2, 6)), axis=0)
            xy = xy[gen.permutation(len(xy))]
            if r == want_round and k == want_slice:
                return xy
    raise ValueError("no such slice")


if __name__ == "__main__":
    xy = np.unique(slice_of(31, 1, 170), axis=0)
    vertices = ConvexHull(xy).vertices.astype(np.int64)
    note = ("a slice of tests/sweep_qhull_device.py (seed 31, round 1, slice 170: 1500 distinct points 7.3e6 from the origin) and "
            "scipy.spatial.ConvexHull(xy).vertices of scipy 1.15.3 in the build container: the exact hull has one vertex more (875)")
    np.savez_compressed(os.path.join(HERE, "qhull_far_vertex.npz"), xy=xy, vertices=vertices, note=np.array(note))
    print(len(xy), "points,", len(vertices), "vertices")
