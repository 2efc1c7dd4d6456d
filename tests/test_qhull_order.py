"""The order of ``scipy.spatial.ConvexHull(xy).vertices`` (circle_fitting.py:189-190: the candidates of the
RANSAC in the order its sampled indices refer to): the Python restatement of Qhull's construction
(oracle.qhull2d) and the library's host-side replay (pcfit_qhull_order_batch, no GPU work) against vertex
lists scipy returned in the build container - tests/golden/qhull_order.npz (840 slices of fourteen kinds) and the
``qhull*`` entries of hull_hybrid.npz and long_slices.npz.  CPU only."""
import hashlib

import numpy as np

from conftest import load_golden
from oracle import qhull2d as Q
from oracle import ref_np as R
from pointcloud_amd import _ffi as F


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _cases():
    g = load_golden("qhull_order")
    kinds = [str(k) for k in g["kinds"]]
    for (kind, n, seed, m), first, count, digest in zip(g["cases"], g["first"], g["count"], g["sha"]):
        xy = Q.make_case(kinds[int(kind)], int(n), int(seed))
        assert len(xy) == int(m)
        yield kinds[int(kind)], xy, int(first), int(count), str(digest)


def test_host_replay_equals_scipy_on_every_recorded_slice():
    cases = list(_cases())
    x = np.concatenate([c[1][:, 0] for c in cases])
    y = np.concatenate([c[1][:, 1] for c in cases])
    off = np.concatenate([[0], np.cumsum([len(c[1]) for c in cases])]).astype(np.int64)
    r = F.qhull_order_batch(x, y, off, n_threads=4)
    unsure = wrong_unsure = 0
    for s, (kind, xy, first, count, digest) in enumerate(cases):
        walk = r["walk"][off[s]:off[s] + r["n_walk"][s]]
        same = r["start"][s] == first and r["n_walk"][s] == count and _sha(walk) == digest
        if r["sure"][s]:
            assert same, (s, kind, len(xy))
        else:
            # only inputs with points exactly in line (where Qhull merges facets) or a roundoff unit apart may
            # be undecided; the replay's answer has been right on all of those so far, too
            assert kind in ("grid", "square", "far", "tiny", "small", "rounded", "lattice"), (s, kind)
            unsure += 1
            wrong_unsure += not same
    assert len(cases) >= 800 and unsure < 0.3 * len(cases) and wrong_unsure == 0
    # general position (every kind but those with points exactly in line): nothing undecided
    general = [s for s, c in enumerate(cases) if c[0] not in ("grid", "square", "rounded", "lattice")]
    assert np.mean(r["sure"][general]) > 0.97


def test_python_restatement_equals_scipy_and_the_host_replay():
    done = 0
    for s, (kind, xy, first, count, digest) in enumerate(_cases()):
        if len(xy) > 2000 or s % 3:
            continue                                        # pure Python: the smaller slices, every third
        got, sure = Q.vertices(xy)
        r = F.qhull_order_batch(xy[:, 0].copy(), xy[:, 1].copy(), None, n_threads=1)
        assert got is not None and np.array_equal(got, r["walk"][:r["n_walk"][0]]) and sure == bool(r["sure"][0]), s
        if sure:
            assert got[0] == first and len(got) == count and _sha(got) == digest, (s, kind)
        done += 1
    assert done > 150


def test_vertex_lists_of_the_reference_runs():
    """hull_hybrid.npz / long_slices.npz hold ConvexHull(xy).vertices of slices the reference itself was run on."""
    g = load_golden("hull_hybrid")
    for i in range(10):
        uniq, q = g[f"uniq{i}"], g[f"qhull{i}"]
        if len(uniq) < 3:
            continue
        r = F.qhull_order_batch(uniq[:, 0].copy(), uniq[:, 1].copy(), None)
        assert r["sure"][0] and np.array_equal(r["walk"][:r["n_walk"][0]], q), i
        got, sure = Q.vertices(uniq)
        assert sure and np.array_equal(got, q), i
    g = load_golden("long_slices")
    for i, (n, seed, outlier, dup) in enumerate(g["cases"]):
        pts = R.make_slab(int(n), int(seed), float(outlier), int(dup))
        keep = (pts[:, 2] >= -0.5) & (pts[:, 2] <= 0.5)
        uniq = R.unique_rows(pts[keep, :2])
        r = F.qhull_order_batch(uniq[:, 0].copy(), uniq[:, 1].copy(), None)
        assert r["sure"][0] and np.array_equal(r["walk"][:r["n_walk"][0]], g[f"qhull{i}"]), i


def test_slices_without_a_hull_and_ragged_batches():
    x = np.array([0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 2.0, 0.0, 1.0, 0.0])
    y = np.array([0.0, 1.0, 0.0, 1.0, 2.0, 0.0, 1.0, 2.0, 0.0, 0.0, 1.0])
    off = np.array([0, 2, 2, 5, 8, 11], np.int64)      # two points | none | one vertical line | one line | a triangle
    r = F.qhull_order_batch(x, y, off)
    assert r["start"].tolist()[:4] == [-1, -1, -1, -1] and r["n_walk"].tolist()[:4] == [0, 0, 0, 0]
    assert r["n_walk"][4] == 3 and sorted(r["walk"][8:11].tolist()) == [0, 1, 2]
    # n_unique shorter than the slice's room: only the first n_unique points count
    r2 = F.qhull_order_batch(x, y, off, n_unique=np.array([2, 0, 3, 3, 2], np.int64))
    assert r2["start"][4] == -1


def test_a_vertex_of_the_exact_hull_that_qhull_never_makes():
    """1500 points 7.3e6 from the origin (tests/golden/qhull_far_vertex.npz: a slice the randomised device sweep drew,
    with scipy's list): point 875 lies 2.7 roundoff units beyond the chord of its neighbours - a vertex of the exact
    hull, none of Qhull's, which asks four units of an outside point.  The host replay and the Python restatement list
    scipy's 80 vertices."""
    g = load_golden("qhull_far_vertex")
    xy, want = g["xy"], g["vertices"]
    host = F.qhull_order_batch(np.ascontiguousarray(xy[:, 0]), np.ascontiguousarray(xy[:, 1]), None)
    assert np.array_equal(host["walk"][:int(host["n_walk"][0])], want)
    verts, _ = Q.vertices(xy)
    assert np.array_equal(np.asarray(verts), want)
