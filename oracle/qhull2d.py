"""The order of ``scipy.spatial.ConvexHull(xy).vertices`` in the plane, restated (TEST INFRASTRUCTURE ONLY).

The reference takes its RANSAC candidates as ``xy[hull.vertices]`` (utils/circle_fitting.py:189-192), so the
vertex the list starts with decides which points the sampled index triples name.  The hull itself comes from a
third-party dependency that is not part of the reference checkout: scipy (1.15.3 in the build container), which
vendors libqhull_r 2020.2 and runs it with triangulated output and Qhull's default merging.  scipy walks Qhull's
facet list from its head (each facet an edge of the hull; per facet the first vertex in the facet's
orientation), so the order follows from the list Qhull's incremental construction leaves behind.  This module
restates that construction for points in general position, from Qhull's published algorithm (quickhull with
outside sets, Barber et al. 1996) and the documented behaviour of its routines named in the comments below:
initial simplex from the coordinate extremes, first partition, farthest-point insertion in facet-list order.
Of Qhull's facet merging only the outcome for points exactly in line on a hull edge is restated (the vertices
between them go, and the merged facet stands in the list where its newest part stood); ``sure`` is False when
a decision fell inside Qhull's roundoff bands, where merging could have changed more than that.

Parity status: pinned by tests/golden/qhull_order.npz and the ``qhull*`` entries of hull_hybrid.npz and
long_slices.npz - vertex lists scipy returned in the build container (tests/golden/gen_golden.py).

python -m oracle.qhull2d [slices]   compares with scipy on random slices (build container only).
"""
from __future__ import annotations

import math
import sys

import numpy as np

EPS = 2.220446049250313e-16


class _Facet:
    __slots__ = ("v", "top", "n", "off", "out", "far", "nb", "visible", "new")

    def __init__(self, v, top):
        self.v, self.top = v, top
        self.out, self.far, self.nb = None, 0.0, [None, None]
        self.visible = self.new = False


def vertices(xy):
    """(vertex indices in scipy's order, sure) for an (n, 2) array of distinct points; (None, False) when
    Qhull could not start (fewer than 3 points, all on one vertical line, no area)."""
    pts = [(float(a), float(b)) for a, b in xy]
    n = len(pts)
    if n < 3:
        return None, False
    sure = [True]
    # qh_maxmin: per coordinate the first point of least / greatest value; qh_distround, qh_detroundoff
    ext, maxabs, sumabs, maxwidth = [], 0.0, 0.0, 0.0
    for k in range(2):
        lo = hi = 0
        for i in range(1, n):
            if pts[i][k] > pts[hi][k]:
                hi = i
            elif pts[i][k] < pts[lo][k]:
                lo = i
        ext += [lo, hi]
        a = max(abs(pts[lo][k]), abs(pts[hi][k]))
        maxabs = max(maxabs, a)
        sumabs += a
        maxwidth = max(maxwidth, pts[hi][k] - pts[lo][k])
    distround = EPS * (2 * min(math.sqrt(2.0) * maxabs, sumabs) * 1.01 + maxabs)
    minvisible = 2 * distround                  # the centrum radius of the default merging
    minoutside = 2 * minvisible
    searchdist = 2 * distround + minvisible
    distoutside = 2 * minoutside
    band = 64 * distround

    def near(d):
        if abs(d) < band:
            sure[0] = False

    # qh_maxsimplex: least and greatest x among the extremes, then the extreme spanning the largest triangle;
    # all other points, in order, if that triangle is flat or its determinant near zero
    minx = maxx = ext[0]
    for i in ext:
        if pts[i][0] > pts[maxx][0]:
            maxx = i
        if pts[i][0] < pts[minx][0]:
            minx = i
    if minx == maxx:
        return None, False

    def det(i):
        a, b, p = pts[minx], pts[maxx], pts[i]
        return abs((a[0] - p[0]) * (b[1] - p[1]) - (b[0] - p[0]) * (a[1] - p[1]))

    third, bestdet = None, -1.0
    for i in ext:
        if i not in (minx, maxx):
            d = det(i)
            if d > bestdet:
                third, bestdet = i, d
    target = (pts[maxx][0] - pts[minx][0]) * maxwidth
    mindet = 10 * 1e-3 * target
    nearzero = 10 * (80 * sumabs * EPS)
    if third is None or bestdet < nearzero or bestdet / target < 1e-3:
        for i in range(n):
            if i in ext:
                continue
            d = det(i)
            if d > bestdet:
                third, bestdet = i, d
                if not d < nearzero and bestdet > mindet:
                    break
        if third is None or not bestdet > 0.0:
            return None, False

    def setplane(f):
        p0, p1 = pts[f.v[0]], pts[f.v[1]]
        nx, ny = p1[1] - p0[1], p0[0] - p1[0]
        norm = math.sqrt(nx * nx + ny * ny)
        if not f.top:
            norm = -norm
        f.n = (nx / norm, ny / norm)
        f.off = -(p0[0] * f.n[0] + p0[1] * f.n[1])

    def dist(p, f):
        q = pts[p]
        return f.off + q[0] * f.n[0] + q[1] * f.n[1]

    # qh_createsimplex, qh_initialhull: vertex list by decreasing age, each facet the list without one vertex,
    # orientation alternating; all flip if the centre lies above the first facet
    vs = [third, maxx, minx]
    facets, top = [], True
    for i in range(3):
        facets.append(_Facet([v for k, v in enumerate(vs) if k != i], top))
        top = not top
    for f in facets:
        f.nb = [g for g in facets if g is not f]
    cx = (pts[minx][0] + pts[maxx][0] + pts[third][0]) / 3.0
    cy = (pts[minx][1] + pts[maxx][1] + pts[third][1]) / 3.0
    setplane(facets[0])
    if facets[0].off + cx * facets[0].n[0] + cy * facets[0].n[1] > 0:
        for f in facets:
            f.top = not f.top
    for f in facets:
        setplane(f)
        if not f.off + cx * f.n[0] + cy * f.n[1] < -band:
            sure[0] = False
    # qh_partitionall: every point to the first facet it lies outside of, the farthest last
    rest = [i for i in range(n) if i not in (minx, maxx, third)]
    for f in facets:
        keep, outs, bestp, bestd = [], [], None, 0.0
        for p in rest:
            d = dist(p, f)
            near(d - distoutside)
            if d < distoutside:
                keep.append(p)
            elif bestp is None:
                bestp, bestd = p, d
            elif d > bestd:
                outs.append(bestp)
                bestp, bestd = p, d
            else:
                outs.append(p)
        if bestp is not None:
            outs.append(bestp)
            f.out, f.far = outs, bestd
        rest = keep
    # the facet list with its tail and qh facet_next, kept as qh_removefacet / qh_appendfacet keep them
    tail = object()
    flist = facets + [tail]
    state = {"next": flist[0], "findbestnew": False, "notsharp": False}
    newlist = []

    def remove(f):
        k0 = flist.index(f)
        if state["next"] is f:
            state["next"] = flist[k0 + 1]
        del flist[k0]

    def append(f):
        if state["next"] is tail:
            state["next"] = f
        flist.insert(len(flist) - 1, f)

    def findbesthorizon(p, start, bestdist):
        best, seen, facet, copl, size, nextfacet = start, {id(start)}, start, [], 0, None
        minsearch = bestdist - searchdist
        while True:
            for nb in facet.nb:
                if id(nb) in seen:
                    continue
                seen.add(id(nb))
                d = dist(p, nb)
                if d > bestdist:
                    minsearch = d - searchdist
                    if d > bestdist + searchdist:
                        size = 0
                    best, bestdist = nb, d
                elif d < minsearch:
                    continue
                if nextfacet is not None:
                    if size == 0:
                        copl, size = [nextfacet], 1
                    else:
                        copl.append(nextfacet)
                        size += 1
                nextfacet = nb
            facet = nextfacet
            if facet is not None:
                nextfacet = None
            elif size == 0:
                break
            elif size == 1:
                size, facet, copl = 0, copl[0], []
            else:
                size -= 1
                facet = copl.pop()
        return best, bestdist

    def findbestnew(p, start):
        best, bestdist = None, -1e300
        k0 = flist.index(start)
        first_new = flist.index(newlist[0]) if newlist else 0
        for f in flist[k0:-1] + flist[first_new:k0]:
            if f.visible:
                continue
            d = dist(p, f)
            near(d - minoutside)
            near(d - distoutside)
            if d > bestdist:
                best = f
                if d >= distoutside:
                    return best, d, True
                bestdist = d
        best, bestdist = findbesthorizon(p, best if best is not None else start, bestdist)
        near(bestdist - minoutside)
        return best, bestdist, bestdist >= minoutside

    def sharpnewfacets():
        a = newlist[0]
        return any((a.n[1] > 0) != (f.n[1] > 0) or (a.n[0] > 0) != (f.n[0] > 0) for f in newlist[1:])

    def findbest(p, start):
        d = dist(p, start)
        near(d - minoutside)
        if d >= minoutside:
            return start, d, True
        best, bestdist, seen, facet = start, d, {id(start)}, start
        while facet is not None:
            nxt = None
            for nb in facet.nb:
                if not nb.new or id(nb) in seen:
                    continue
                seen.add(id(nb))
                d = dist(p, nb)
                near(d - minoutside)
                if d > bestdist:
                    if d >= minoutside:
                        return nb, d, True
                    best, bestdist, nxt = nb, d, nb
                    break
            facet = nxt
        testhorizon = True
        if not state["notsharp"] and bestdist < -distround:
            if sharpnewfacets():
                best = findbestnew(p, best)[0]
                testhorizon = False
                state["findbestnew"] = True
            else:
                state["notsharp"] = True
        if testhorizon:
            best, bestdist = findbesthorizon(p, best, bestdist)
        near(bestdist - minoutside)
        return best, bestdist, bestdist >= minoutside

    def partitionpoint(p, start):
        f, d, isout = findbestnew(p, start) if state["findbestnew"] else findbest(p, start)
        if not isout:
            return
        if not f.out:
            f.out = [p]
            if not f.new:
                remove(f)
                append(f)
            f.far = d
        elif f.far < d:
            f.out.append(p)
            f.far = d
        else:
            f.out.insert(len(f.out) - 1, p)

    # the rest of qh_partitionall: what no facet took, through qh_findbestnew
    for f in facets:
        f.new = True
    newlist = list(facets)
    state["findbestnew"] = True
    for p in rest:
        partitionpoint(p, flist[0])
    state["findbestnew"] = False
    for f in facets:
        f.new = False
    # qh_furthestnext: the facet with the farthest outside point goes to the front of the list
    bestf, bestd = None, -1e300
    for f in flist[:-1]:
        if f.out and f.far > bestd:
            bestf, bestd = f, f.far
    if bestf is not None:
        flist.remove(bestf)
        flist.insert(0, bestf)
    state["next"] = flist[0]
    # qh_buildhull
    while True:
        while state["next"] is not tail and not state["next"].out:             # qh_nextfurthest
            state["next"] = flist[flist.index(state["next"]) + 1]
        if state["next"] is tail:
            break
        facet = state["next"]
        apex = facet.out.pop()
        # qh_findhorizon: visible facets move to the end of the list in the order they are found
        remove(facet)
        append(facet)
        facet.visible = True
        vis, seen, k = [facet], {id(facet)}, 0
        while k < len(vis):
            for g in vis[k].nb:
                if id(g) in seen:
                    continue
                seen.add(id(g))
                d = dist(apex, g)
                near(d - minvisible)
                if d > minvisible:
                    remove(g)
                    append(g)
                    g.visible = True
                    vis.append(g)
                elif d >= -band:
                    sure[0] = False
            k += 1
        # qh_makenew_simplicial: one facet per horizon ridge, the apex first
        new = []
        for v in vis:
            for g in v.nb:
                if g.visible:
                    continue
                hskip = g.nb.index(v)
                f = _Facet([apex, g.v[1 - hskip]], bool(hskip & 1) if g.top else not bool(hskip & 1))
                f.new = True
                f.nb = [g, None]
                g.nb[hskip] = f
                setplane(f)
                append(f)
                new.append(f)
        if len(new) != 2:
            return None, False
        new[0].nb[1], new[1].nb[1] = new[1], new[0]
        newlist = new
        # where the default merging would look twice: a new facet nearly in line with a neighbour
        for a, b in ((new[0], new[1]), (new[1], new[0])):
            g = a.nb[0]
            w = g.v[1] if g.v[0] == a.v[1] else g.v[0]
            if dist(w, a) > -band or dist(apex, g) > -band or dist(b.v[1], a) > -band:
                sure[0] = False
        # qh_partitionvisible, qh_deletevisible
        state["findbestnew"] = state["notsharp"] = False
        for v in vis:
            outs, v.out = v.out or [], None
            for p in outs:
                partitionpoint(p, new[0])
        state["findbestnew"] = state["notsharp"] = False
        for v in vis:
            remove(v)
        for f in new:
            f.new = False
    # scipy's walk: from the head facet, each facet's first vertex in its orientation, on to the neighbour
    # opposite that vertex.  Facets in one line (points exactly on a hull edge) are one facet in Qhull's result,
    # which merged them: the vertex between two facets is left out when the centrum of one lies on the line of
    # the other (Qhull's test), and the merged facet stands in the list where its newest part stands.
    walk, f = [], flist[0]
    for _ in range(len(flist) - 1):
        walk.append(f)
        f = f.nb[0] if f.top else f.nb[1]
        if f is flist[0]:
            break
    m = len(walk)

    def centrum_dist(a, b):             # a's centrum (the mean of its vertices, brought onto its own line) from b's line
        cx = (pts[a.v[0]][0] + pts[a.v[1]][0]) / 2
        cy = (pts[a.v[0]][1] + pts[a.v[1]][1]) / 2
        own = a.off + cx * a.n[0] + cy * a.n[1]
        cx -= own * a.n[0]
        cy -= own * a.n[1]
        return b.off + cx * b.n[0] + cy * b.n[1]

    def in_line(i):                     # Qhull's test of two facets: a centrum within the centrum radius of the other's line
        radius = 2 * distround
        return abs(centrum_dist(walk[i], walk[i - 1])) <= radius or abs(centrum_dist(walk[i - 1], walk[i])) <= radius

    joined = [in_line(i) for i in range(m)]
    if all(joined) or m - sum(joined) < 3:
        joined = [False] * m            # (no area left: nothing sensible to merge)
    # runs of facets in one line, as (first facet's index in the walk, place in the list); a merged facet was
    # appended to the list when its newest part was made, so it stands where that part stands
    place = {id(f): k for k, f in enumerate(flist[:-1])}
    first = next(i for i in range(m) if not joined[i])
    runs, i = [], first
    for _ in range(m):
        if not joined[i]:
            runs.append([i, place[id(walk[i])]])
        else:
            runs[-1][1] = max(runs[-1][1], place[id(walk[i])])
        i = (i + 1) % m
    head = min(range(len(runs)), key=lambda k: runs[k][1])
    runs = runs[head:] + runs[:head]
    order = [walk[i].v[0] if walk[i].top else walk[i].v[1] for i, _ in runs]
    return np.array(order, dtype=np.int64), sure[0]


# --------------------------------------------------------------------------
# inputs of the qhull_order fixtures (tests/golden/gen_golden.py records scipy's answer for them)
# --------------------------------------------------------------------------
KINDS = ("ring", "disk", "gauss", "grid", "far", "tiny", "square", "ellipse", "ringf32", "small", "arc", "slab",
         "rounded", "lattice")


def make_case(kind, n, seed):
    """Sorted distinct points of one fixture slice."""
    g = np.random.default_rng(seed)
    th = g.uniform(0, 2 * np.pi, n)
    if kind == "ring":
        r = 7 + g.normal(0, 0.02, n)
        xy = np.stack([r * np.cos(th) + 1, r * np.sin(th) - 3], 1)
    elif kind == "disk":
        r = 5 * np.sqrt(g.random(n))
        xy = np.stack([r * np.cos(th), r * np.sin(th)], 1)
    elif kind == "gauss":
        xy = g.normal(0, 1, (n, 2))
    elif kind == "grid":
        xy = g.integers(0, 12, (n, 2)).astype(float)
    elif kind == "far":
        r = 2 + g.normal(0, 0.01, n)
        xy = np.stack([r * np.cos(th) + 4.5e5, r * np.sin(th) - 7.3e6], 1)
    elif kind == "tiny":
        r = 1e-6 * (1 + g.normal(0, 0.01, n))
        xy = np.stack([r * np.cos(th) + 1, r * np.sin(th)], 1)
    elif kind == "square":
        xy = g.uniform(-1, 1, (n, 2))
        xy[g.random(n) < 0.3, 0] = 1.0
    elif kind == "ellipse":
        xy = np.stack([30 * np.cos(th), 0.01 * np.sin(th)], 1) + g.normal(0, 0.003, (n, 2))
    elif kind == "ringf32":
        r = 7 + g.normal(0, 0.02, n)
        xy = np.stack([r * np.cos(th) + 1, r * np.sin(th) - 3], 1).astype(np.float32).astype(float)
    elif kind == "small":
        xy = g.normal(0, 1, (int(g.integers(3, 7)), 2))
    elif kind == "arc":
        r = 7 + g.normal(0, 0.02, n)
        xy = np.stack([r * np.cos(0.2 * th), r * np.sin(0.2 * th)], 1)
    elif kind == "slab":
        r = 10 + g.normal(0, 0.05, n)
        bad = g.random(n) < 0.3
        r[bad] = g.uniform(0, 14.0, int(bad.sum()))
        xy = np.stack([3 + r * np.cos(th), -2 + r * np.sin(th)], 1)
    elif kind == "rounded":         # a ring as a text file with few decimals holds it: points exactly in line
        r = 10 + g.normal(0, 0.005, n)
        xy = np.round(np.stack([3 + r * np.cos(th), -2 + r * np.sin(th)], 1), int(g.integers(0, 4)))
    elif kind == "lattice":         # a handful of points on a small integer lattice
        xy = g.integers(0, int(g.choice([4, 6, 9, 15])), (int(g.integers(4, 14)), 2)).astype(float)
    else:
        raise ValueError(kind)
    return np.unique(xy, axis=0)


if __name__ == "__main__":
    from scipy.spatial import ConvexHull
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    gen = np.random.default_rng(1)
    tally = {}
    for r in range(rounds):
        kind = KINDS[r % len(KINDS)]
        xy = make_case(kind, int(gen.choice([4, 10, 30, 200, 1500])), int(gen.integers(1 << 30)))
        if len(xy) < 3:
            continue
        try:
            want = ConvexHull(xy).vertices
        except Exception:           # noqa: BLE001  (Qhull refuses flat input)
            continue
        got, ok = vertices(xy)
        t = tally.setdefault(kind, [0, 0, 0, 0])
        same = got is not None and np.array_equal(got, want)
        t[0] += 1
        t[1] += ok and not same
        t[2] += not ok
        t[3] += (not ok) and not same
    print("kind: slices, wrong while sure, unsure, wrong while unsure")
    for kind, t in tally.items():
        print(f"{kind:8s} {t}")
