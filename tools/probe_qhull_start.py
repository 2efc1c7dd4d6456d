"""Which vertex does scipy's ConvexHull(xy).vertices start with?  (build container only: needs scipy)
Probe over random slices of the synthetic cylinder: the first vertex against simple candidate rules."""
import sys
import numpy as np
from scipy.spatial import ConvexHull

sys.path.insert(0, ".")
from oracle import ref_np as R

rules = {}


def rule(f):
    rules[f.__name__] = f
    return f


def tri_area2(p, q, r):
    return abs((q[0] - p[0]) * (r[1] - p[1]) - (q[1] - p[1]) * (r[0] - p[0]))


def simplex(xy):
    """Qhull's initial simplex in 2-D as its sources describe it: the points of least and greatest x, then,
    among the extreme points of the coordinates, the one spanning the largest triangle with them."""
    imin, imax = int(np.argmin(xy[:, 0])), int(np.argmax(xy[:, 0]))
    cands = [int(np.argmin(xy[:, 1])), int(np.argmax(xy[:, 1]))]
    third = max(cands, key=lambda k: tri_area2(xy[imin], xy[imax], xy[k]))
    return imin, imax, third


@rule
def lexicographic_min(xy, hull):
    return 0


@rule
def max_x(xy, hull):
    return int(np.argmax(xy[:, 0]))


@rule
def min_y(xy, hull):
    return int(np.argmin(xy[:, 1]))


@rule
def max_y(xy, hull):
    return int(np.argmax(xy[:, 1]))


@rule
def third_simplex_vertex(xy, hull):
    return simplex(xy)[2]


@rule
def any_simplex_vertex(xy, hull):
    s = simplex(xy)
    return hull[0] if hull[0] in s else -1


@rule
def farthest_from_minmax_chord(xy, hull):
    imin, imax, _ = simplex(xy)
    d = [tri_area2(xy[imin], xy[imax], p) for p in xy]
    return int(np.argmax(d))


@rule
def ccw_successor_of_a_simplex_vertex(xy, hull):
    s = simplex(xy)
    pred = hull[-1]
    return hull[0] if pred in s else -1


def main(n_slices=300):
    gen = np.random.default_rng(0)
    hits = {k: 0 for k in rules}
    pos_hist = []
    for k in range(n_slices):
        n = int(gen.choice([300, 2000, 5000]))
        outl = float(gen.choice([0.0, 0.02, 0.3]))
        xy = R.unique_rows(R.make_slab(n, 5000 + k, outlier=outl)[:, :2])
        hull = ConvexHull(xy).vertices
        mine = R.hull_ccw(xy)
        assert set(hull) == set(mine)
        pos_hist.append(int(np.nonzero(mine == hull[0])[0][0]) / len(mine))
        for name, f in rules.items():
            hits[name] += int(f(xy, list(hull)) == hull[0])
    for name, h in hits.items():
        print(f"{name:40s} {h:4d} / {n_slices}")
    print("position of Qhull's first vertex along the walk from the lexicographic minimum (deciles):",
          np.histogram(pos_hist, bins=10, range=(0, 1))[0].tolist())


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 300)
