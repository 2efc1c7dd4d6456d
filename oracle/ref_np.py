"""numpy restatement of the reference hot path (TEST INFRASTRUCTURE ONLY).

Every function names the reference lines it restates (paths relative to the
reference checkout).  The arithmetic is kept on the same numpy / LAPACK entry
points the reference uses (``np.sum`` pairwise sums, ``np.linalg.solve`` for
2x2 and 3x3 systems, ``np.hypot`` / ``np.arctan2``, ``default_rng().choice``),
because those define the reference's bits.  Only the convex hull differs: the
reference calls Qhull through scipy, this file uses a monotone chain with an
exact orientation test and starts the counter-clockwise walk at the
lexicographically smallest vertex; where Qhull's own start vertex is wanted
(``hull_start="qhull"``) it comes from oracle/qhull2d.py, which restates
Qhull's construction.

Parity status: pinned by tests/golden/*.npz, which hold outputs of the real
reference recorded by tests/golden/gen_golden.py in the build container.
"""
from __future__ import annotations

from fractions import Fraction

import numpy as np

TWO_PI = 2 * np.pi


# --------------------------------------------------------------------------
# slicing                                                  analyzer.py:49-52, 232-243
# --------------------------------------------------------------------------
def slice_windows(z_min: float, z_max: float, z_step: float):
    """Window centres and closed bounds.  analyzer.py:240-243 and :49-50."""
    half = z_step / 2.0
    centres = np.arange(z_min + half, z_max - half + 1e-9, z_step)
    return centres, centres - half, centres + half


def slice_members(z: np.ndarray, lo: float, hi: float) -> np.ndarray:
    """Indices (ascending = original order) of points in [lo, hi].  analyzer.py:50."""
    return np.nonzero((z >= lo) & (z <= hi))[0]


def bin_slices(z: np.ndarray, lo: np.ndarray, hi: np.ndarray):
    """CSR form of the per-slice masks: offsets (S+1) and the concatenated,
    order-preserving member indices.  One closed-interval mask per window,
    exactly as analyzer.py:50 evaluates it."""
    members = [slice_members(z, a, b) for a, b in zip(lo, hi)]
    counts = np.array([m.size for m in members], dtype=np.int64)
    offsets = np.zeros(len(members) + 1, dtype=np.int64)
    np.cumsum(counts, out=offsets[1:])
    perm = np.concatenate(members) if members else np.zeros(0, np.int64)
    return offsets, perm.astype(np.int64)


# --------------------------------------------------------------------------
# small helpers                                            circle_fitting.py:14-32
# --------------------------------------------------------------------------
def has_2d_variation(xy: np.ndarray, eps: float = 1e-12) -> bool:
    """Second singular value of the centred cloud against the first.
    circle_fitting.py:14-19."""
    if xy.shape[0] < 3:
        return False
    centred = xy - xy.mean(axis=0, keepdims=True)
    sv = np.linalg.svd(centred, full_matrices=False)[1]
    return bool(sv[1] > eps * (sv[0] + eps))


def circle_3pt(p1, p2, p3):
    """Circumcircle by intersecting two perpendicular bisectors; None when the
    2x2 determinant is below 1e-12 in magnitude.  circle_fitting.py:22-32."""
    (x1, y1), (x2, y2), (x3, y3) = p1, p2, p3
    a = 2 * (x2 - x1)
    b = 2 * (y2 - y1)
    c = x2 * x2 + y2 * y2 - x1 * x1 - y1 * y1
    d = 2 * (x3 - x2)
    e = 2 * (y3 - y2)
    f = x3 * x3 + y3 * y3 - x2 * x2 - y2 * y2
    lhs = np.array([[a, b], [d, e]], float)
    if abs(np.linalg.det(lhs)) < 1e-12:
        return None
    cx, cy = np.linalg.solve(lhs, np.array([c, f], float))
    return float(cx), float(cy), float(np.hypot(x1 - cx, y1 - cy))


# --------------------------------------------------------------------------
# algebraic fit                                            circle_fitting.py:35-54
# --------------------------------------------------------------------------
def kasa_fit(x, y):
    """The reference's ``fit_circle_pratt`` (a Kasa fit): centre the data on its
    mean, seven centred sums, 2x2 normal equations.  circle_fitting.py:35-54."""
    x = np.asarray(x, float).ravel()
    y = np.asarray(y, float).ravel()
    n = x.size
    if n < 3:
        raise ValueError("Need at least 3 points for Pratt fit")
    mx, my = x.mean(), y.mean()
    u, v = x - mx, y - my
    suu, svv, suv = np.sum(u * u), np.sum(v * v), np.sum(u * v)
    suuu, svvv = np.sum(u * u * u), np.sum(v * v * v)
    suvv, svuu = np.sum(u * v * v), np.sum(v * u * u)
    lhs = np.array([[2 * suu, 2 * suv], [2 * suv, 2 * svv]], float)
    rhs = np.array([suuu + suvv, svvv + svuu], float)
    try:
        uc, vc = np.linalg.solve(lhs, rhs)
    except np.linalg.LinAlgError:
        uc, vc = np.linalg.lstsq(lhs, rhs, rcond=None)[0]
    return float(mx + uc), float(my + vc), float(np.sqrt(uc * uc + vc * vc + (suu + svv) / n))


# --------------------------------------------------------------------------
# geometric refinement                                     circle_fitting.py:57-109
# --------------------------------------------------------------------------
def lm_huber(x, y, cx0, cy0, r0, huber_delta=1.0, max_iter=100, lm_lambda0=1e-2,
             tol_update=1e-10, tol_cost=1e-12, trace=None):
    """Levenberg-Marquardt on the geometric residual with Huber IRLS weights.
    circle_fitting.py:57-109.  ``trace`` (a list) receives one tuple per
    iteration: (accepted, lambda_used, cost_candidate, cx, cy, R after it)."""
    x = np.asarray(x, float).ravel()
    y = np.asarray(y, float).ravel()
    cx, cy, rad = float(cx0), float(cy0), float(r0)

    def radial(px, py, pr):
        ex, ey = x - px, y - py
        dist = np.sqrt(ex * ex + ey * ey) + 1e-16
        return dist - pr, ex, ey, dist

    lam = lm_lambda0
    best_cost = np.inf
    for _ in range(max_iter):
        res, ex, ey, dist = radial(cx, cy, rad)
        mag = np.abs(res)
        wts = np.ones_like(res)
        far = mag > huber_delta
        wts[far] = huber_delta / (mag[far] + 1e-16)
        jac = np.stack([-ex / dist, -ey / dist, -np.ones_like(dist)], axis=1)
        root = np.sqrt(wts)
        jw = jac * root[:, None]
        rw = res * root
        hess = jw.T @ jw
        grad = jw.T @ rw
        damped = hess + lam * np.diag(np.diag(hess))
        lam_used = lam
        try:
            step = -np.linalg.solve(damped, grad)
        except np.linalg.LinAlgError:
            lam *= 3
            if trace is not None:
                trace.append((-1, lam_used, np.nan, cx, cy, rad))
            continue
        tx, ty, tr = cx + step[0], cy + step[1], max(1e-12, rad + step[2])
        mag_new = np.abs(radial(tx, ty, tr)[0])
        near = mag_new <= huber_delta
        cost = 0.5 * np.sum(mag_new[near] ** 2) + np.sum(huber_delta * (mag_new[~near] - 0.5 * huber_delta))
        accepted = bool(cost < best_cost - tol_cost)
        stop = False
        if accepted:
            cx, cy, rad = tx, ty, tr
            best_cost = cost
            lam = max(lam / 3, 1e-12)
            stop = bool(np.linalg.norm(step) < tol_update)
        else:
            lam *= 3
        if trace is not None:
            trace.append((int(accepted), lam_used, float(cost), float(cx), float(cy), float(rad)))
        if stop:
            break
    return float(cx), float(cy), float(rad)


# --------------------------------------------------------------------------
# RANSAC                                                   circle_fitting.py:112-143
# --------------------------------------------------------------------------
def ransac(xy, inlier_tol=0.5, max_trials=4000, min_inliers_frac=0.1, r_range=None,
           rng_seed=42, want_margin=False):
    """Seeded 3-point RANSAC.  The triple is drawn before any rejection so the
    trial -> triple map depends on (seed, n) alone; the earliest trial with the
    largest consensus wins (strict '>').  circle_fitting.py:112-143.

    Returns None or a dict with score, cx, cy, R, mask, best_trial (and, with
    ``want_margin``, the smallest |err - tol| seen over every scored trial)."""
    xy = np.asarray(xy, float)
    n = xy.shape[0]
    if n < 3:
        return None
    best = dict(score=-1, cx=None, cy=None, R=None, mask=None, best_trial=-1)
    margin = np.inf
    gen = np.random.default_rng(rng_seed)
    for trial in range(max_trials):
        pick = gen.choice(n, size=3, replace=False)
        circ = circle_3pt(xy[pick[0]], xy[pick[1]], xy[pick[2]])
        if circ is None:
            continue
        cx, cy, rad = circ
        if r_range is not None and not (r_range[0] <= rad <= r_range[1]):
            continue
        ex = xy[:, 0] - cx
        ey = xy[:, 1] - cy
        err = np.abs(np.sqrt(ex * ex + ey * ey) - rad)
        inl = err <= inlier_tol
        if want_margin:
            margin = min(margin, float(np.min(np.abs(err - inlier_tol))))
        score = int(inl.sum())
        if score > best["score"]:
            best.update(score=score, cx=cx, cy=cy, R=rad, mask=inl, best_trial=trial)
    if best["score"] < max(int(min_inliers_frac * n), 3):
        return None
    if want_margin:
        best["margin"] = margin
    return best


def choice3_table(n: int, trials: int, seed=42) -> np.ndarray:
    """The (trials, 3) index triples ``default_rng(seed).choice(n, 3, replace=False)``
    yields call after call.  circle_fitting.py:124-126."""
    gen = np.random.default_rng(seed)
    return np.array([gen.choice(n, size=3, replace=False) for _ in range(trials)], dtype=np.int64)


# --------------------------------------------------------------------------
# unique + convex hull                                     circle_fitting.py:182-192
# --------------------------------------------------------------------------
def unique_rows(xy: np.ndarray) -> np.ndarray:
    """Lexicographically sorted distinct rows.  circle_fitting.py:183."""
    return np.unique(np.asarray(xy, float), axis=0)


def _orient_sign(a, b, c) -> int:
    """Sign of (b-a) x (c-a), exact.  A floating-point filter decides almost
    every call; rational arithmetic on the doubles settles the rest."""
    l = (b[0] - a[0]) * (c[1] - a[1])
    r = (b[1] - a[1]) * (c[0] - a[0])
    det = l - r
    bound = 1e-14 * (abs(l) + abs(r))
    if det > bound:
        return 1
    if det < -bound:
        return -1
    ax, ay, bx, by, cx, cy = (Fraction(float(t)) for t in (a[0], a[1], b[0], b[1], c[0], c[1]))
    exact = (bx - ax) * (cy - ay) - (by - ay) * (cx - ax)
    return (exact > 0) - (exact < 0)


def hull_ccw(pts_sorted: np.ndarray) -> np.ndarray:
    """Indices into a lexicographically sorted, duplicate-free (n,2) array of
    the strictly convex hull vertices, counter-clockwise, starting at index 0
    (the lexicographic minimum).  Stands in for ``ConvexHull(xy).vertices``
    (circle_fitting.py:189-190): same vertex set and cyclic order, canonical
    start."""
    n = pts_sorted.shape[0]
    if n < 3:
        return np.arange(n, dtype=np.int64)
    p = pts_sorted
    chain: list[int] = []
    for i in range(n):                              # lower chain, left to right
        while len(chain) >= 2 and _orient_sign(p[chain[-2]], p[chain[-1]], p[i]) <= 0:
            chain.pop()
        chain.append(i)
    floor = len(chain) + 1
    for i in range(n - 2, -1, -1):                  # upper chain, right to left
        while len(chain) >= floor and _orient_sign(p[chain[-2]], p[chain[-1]], p[i]) <= 0:
            chain.pop()
        chain.append(i)
    chain.pop()                                     # index 0 closes the loop
    return np.array(chain, dtype=np.int64)


def rotate_to(seq: np.ndarray, first) -> np.ndarray:
    """Cyclic rotation of ``seq`` that begins at the element equal to ``first``."""
    k = int(np.nonzero(seq == first)[0][0])
    return np.concatenate([seq[k:], seq[:k]])


# --------------------------------------------------------------------------
# hybrid driver                                            circle_fitting.py:146-232
# --------------------------------------------------------------------------
def hybrid_fit(points_xyz, z_elevation, z_tolerance, inlier_tol=0.5, max_trials=4000,
               min_inliers_frac=0.1, r_range=None, huber_delta=1.0,
               use_hull_vertices_only=True, hull_start=None):
    """mask -> unique -> rank check -> hull -> RANSAC -> Kasa -> LM.
    circle_fitting.py:146-232.  ``hull_start`` (an index into the unique array,
    or "qhull" for the vertex ``ConvexHull(xy).vertices`` lists first) rotates
    the candidate order."""
    points_xyz = np.asarray(points_xyz)
    if points_xyz.ndim != 2 or points_xyz.shape[1] != 3:
        raise ValueError("points_xyz must be (N,3)")
    z = points_xyz[:, 2]
    keep = (z >= z_elevation - z_tolerance) & (z <= z_elevation + z_tolerance)
    if not np.any(keep):
        return None
    xy = points_xyz[keep, :2]
    if xy.shape[0] < 3:
        return None
    xy = unique_rows(xy)
    cand = None
    if xy.shape[0] >= 3 and has_2d_variation(xy):
        verts = hull_ccw(xy)
        if isinstance(hull_start, str):
            from . import qhull2d
            order, _ = qhull2d.vertices(xy)
            if order is not None:
                verts = order                   # Qhull's own list: its tolerance may leave out a vertex of the exact hull
        elif hull_start is not None:
            verts = rotate_to(verts, hull_start)
        if use_hull_vertices_only and verts.size >= 3:
            cand = xy[verts]
    if cand is None:
        cand = xy
    if cand.shape[0] < 3 or not has_2d_variation(cand):
        return None
    model = ransac(cand, inlier_tol=inlier_tol, max_trials=max_trials,
                   min_inliers_frac=min_inliers_frac, r_range=r_range)
    if model is None:
        return None
    inl = cand[model["mask"]]
    if inl.shape[0] >= 3:
        start = kasa_fit(inl[:, 0], inl[:, 1])
    else:
        start = (model["cx"], model["cy"], model["R"])
    cx, cy, rad = lm_huber(inl[:, 0], inl[:, 1], *start, huber_delta=huber_delta)
    return {
        "center_x": cx, "center_y": cy, "radius": rad,
        "inlier_mask": model["mask"], "fit_points_xy": xy, "candidates_xy": cand,
        "z_elevation": float(z_elevation),
    }


# --------------------------------------------------------------------------
# boundaries, ovality, residuals                           boundary.py, gui.py:972-987
# --------------------------------------------------------------------------
def polar(xy, cx, cy):
    """theta in [0, 2pi) and r = hypot.  boundary.py:19-22, analyzer.py:83-86."""
    ex = xy[:, 0] - cx
    ey = xy[:, 1] - cy
    return (np.arctan2(ey, ex) + TWO_PI) % TWO_PI, np.hypot(ex, ey)


def angle_max_indices(xy, center=None, bins=720):
    """Per non-empty angular bin, the index of the first point of maximal
    radius, then ordered by that point's angle.  boundary.py:7-42 (which
    returns the coordinates; indices are what is compared bit for bit)."""
    xy = np.asarray(xy, float)
    if center is None:
        cx, cy, _ = kasa_fit(xy[:, 0], xy[:, 1])
    else:
        cx, cy = center
    theta, r = polar(xy, cx, cy)
    which = np.clip(np.floor(theta / (TWO_PI / bins)).astype(int), 0, bins - 1)
    picked = []
    for b in range(bins):
        idx = np.nonzero(which == b)[0]
        if idx.size:
            picked.append(idx[np.argmax(r[idx])])
    if not picked:
        raise RuntimeError("No boundary points extracted (data too sparse?)")
    picked = np.array(picked, dtype=np.int64)
    return picked[np.argsort(theta[picked])]


def boundary_angle_max(xy, center=None, bins=720):
    """boundary.py:7-42."""
    xy = np.asarray(xy, float)
    return xy[angle_max_indices(xy, center, bins)]


def boundary_hull_f32(xy):
    """Hull of the float32-rounded points, returned as float64; RuntimeError
    below 10 vertices.  boundary.py:44-54 (OpenCV's hull is not available; the
    vertex *set* is what this restates, ordered counter-clockwise from the
    lexicographic minimum)."""
    p32 = unique_rows(np.asarray(xy).astype(np.float32).astype(np.float64).reshape(-1, 2))
    edge = p32[hull_ccw(p32)]
    if len(edge) < 10:
        raise RuntimeError("Too few hull vertices")
    return edge


def ovality_fourier(x, y, center_x, center_y, radius):
    """Rmax/Rmin/ovality and the k=2 Fourier amplitude of the deviation profile
    resampled on max(1024, n) uniform angles.  boundary.py:56-90."""
    theta = (np.arctan2(y - center_y, x - center_x) + TWO_PI) % TWO_PI
    r = np.hypot(x - center_x, y - center_y)
    dev = r - radius
    order = np.argsort(theta)
    theta, dev, r = theta[order], dev[order], r[order]
    r_hi, r_lo = r.max(), r.min()
    spread = r_hi - r_lo
    pct = 100.0 * spread / radius if radius > 0 else np.nan
    count = max(1024, len(dev))
    grid = np.linspace(0, TWO_PI, count, endpoint=False)
    dev_grid = np.interp(grid, theta, dev, period=TWO_PI)
    amps = np.abs(np.fft.rfft(dev_grid) / (count / 2))
    return {
        "Rmax": r_hi, "Rmin": r_lo, "ovality_abs": spread, "ovality_pct": pct,
        "k2_amp": amps[2] if len(amps) > 2 else 0.0, "theta": grid, "dev_uniform": dev_grid,
    }


def residual_profile(xy, cx, cy, rad):
    """theta, delta_r = hypot - R, both ordered by theta.  gui.py:972-987
    (1.py:331-343 applies the same to every slice point)."""
    theta, r = polar(np.asarray(xy, float), cx, cy)
    order = np.argsort(theta)
    return theta[order], (r - rad)[order]


def filtered_residual_profile(xy, cx, cy, rad, inlier_mask=None):
    """gui.py:1626-1655 without the plotting: delta_r of the edge points filtered by the inlier mask (when it has
    one entry per point) or by the 95 % quantile of |delta_r|, ordered by theta, theta in degrees."""
    xy = np.asarray(xy, float)
    dx = xy[:, 0] - cx
    dy = xy[:, 1] - cy
    theta = (np.arctan2(dy, dx) + 2 * np.pi) % (2 * np.pi)
    delta_r = np.hypot(dx, dy) - rad
    if inlier_mask is not None and len(inlier_mask) == len(xy):
        valid = np.asarray(inlier_mask, bool)
    else:
        mag = np.abs(delta_r)
        valid = mag <= np.quantile(mag, 0.95)
    theta, delta_r = theta[valid], delta_r[valid]
    order = np.argsort(theta)
    return np.degrees(theta[order]), delta_r[order], valid


def residual_profile_at_z(points_xyz, res_fit, z_elevation, z_tolerance, use_hull_vertices=True, hull_start=None):
    """Profile delta_r(theta) of one z window about a fitted circle, over the hull
    vertices of the window's distinct points or over all of them.  1.py:301-345.
    ``hull_start`` replays Qhull's first vertex (the walk otherwise starts at the
    lexicographic minimum; after the ordering by theta it only matters for ties)."""
    cx, cy, rad = float(res_fit["center_x"]), float(res_fit["center_y"]), float(res_fit["radius"])
    z = points_xyz[:, 2]
    keep = (z >= z_elevation - z_tolerance) & (z <= z_elevation + z_tolerance)
    if not np.any(keep):
        return None
    xy = points_xyz[keep, :2].astype(float, copy=False)
    if xy.shape[0] < 3:
        return None
    xy = unique_rows(xy)
    used = xy
    if use_hull_vertices and has_2d_variation(xy):
        verts = hull_ccw(xy)
        if isinstance(hull_start, str):
            from . import qhull2d
            order, _ = qhull2d.vertices(xy)
            if order is not None:
                verts = order                   # Qhull's own list: its tolerance may leave out a vertex of the exact hull
        elif hull_start is not None:
            verts = rotate_to(verts, hull_start)
        if verts.size >= 3:
            used = xy[verts]
    ex = used[:, 0] - cx
    ey = used[:, 1] - cy
    theta = (np.arctan2(ey, ex) + 2 * np.pi) % (2 * np.pi)
    delta_r = np.hypot(ex, ey) - rad
    order = np.argsort(theta)
    return {"theta": theta[order], "delta_r": delta_r[order], "xy_used": used[order], "cx": cx, "cy": cy, "R": rad}


def make_slab(n, seed, outlier=0.0, dup=0, sigma=0.05, half_height=0.9):
    """(n + dup, 3) points of a noisy ring (radius 10 about (3, -2)) with z uniform in +-half_height;
    ``dup`` of the points appear twice.  Input generator of the long-slice fixtures."""
    g = np.random.default_rng(seed)
    th = g.uniform(0.0, 2 * np.pi, n)
    rr = 10.0 + g.normal(0, sigma, n)
    if outlier > 0:
        bad = g.random(n) < outlier
        rr[bad] = g.uniform(0, 14.0, int(bad.sum()))
    xy = np.stack([3.0 + rr * np.cos(th), -2.0 + rr * np.sin(th)], axis=1)
    if dup:
        xy = np.concatenate([xy, xy[g.integers(0, n, dup)]])
    z = g.uniform(-half_height, half_height, len(xy))
    return np.column_stack([xy, z])


def residual_histogram(delta_r, bins, lo, hi):
    """Counts of delta_r over ``bins`` equal bins on [lo, hi]: half-open bins,
    last bin closed, values outside dropped (numpy.histogram semantics).  Not in
    the reference; defined by the build's target statement."""
    return np.histogram(delta_r, bins=bins, range=(lo, hi))[0].astype(np.int64)


# --------------------------------------------------------------------------
# per-slice pipeline and batch driver                      analyzer.py:46-129, 181-230
# --------------------------------------------------------------------------
def process_slice(points, z_center, z_step=2.0, min_points_per_slice=1000,
                  max_points_for_speed=1_000_000, hull_start=None, max_trials=4000):
    """analyzer.py:46-129 without the pandas frame: returns (result, edge, warn)
    where ``edge`` is a dict of the edge-record columns."""
    half = z_step / 2.0
    keep = (points[:, 2] >= z_center - half) & (points[:, 2] <= z_center + half)
    n0 = int(np.count_nonzero(keep))
    if n0 < min_points_per_slice:
        return None, None, f"SKIP zc={z_center:.3f}: too sparse ({n0} points)"
    n0 = min(n0, max_points_for_speed)
    fit = hybrid_fit(points[keep], z_center, half, inlier_tol=0.5, max_trials=max_trials,
                     min_inliers_frac=0.1, huber_delta=1.0, use_hull_vertices_only=True,
                     hull_start=hull_start)
    if fit is None:
        return None, None, f"SKIP zc={z_center:.3f}: hybrid fit failed"
    cx, cy, rad = fit["center_x"], fit["center_y"], fit["radius"]
    cand = fit["candidates_xy"]
    theta, r = polar(cand, cx, cy)
    i_hi, i_lo = int(np.argmax(r)), int(np.argmin(r))
    ov = ovality_fourier(cand[:, 0], cand[:, 1], cx, cy, rad)
    result = {
        "z_center": z_center, "z_low": z_center - half, "z_high": z_center + half,
        "n_points_slice": n0, "n_edge": int(len(cand)),
        "cx": cx, "cy": cy, "R": rad, "Rmax": ov["Rmax"], "Rmin": ov["Rmin"],
        "x_at_Rmax": float(cand[i_hi, 0]), "y_at_Rmax": float(cand[i_hi, 1]),
        "x_at_Rmin": float(cand[i_lo, 0]), "y_at_Rmin": float(cand[i_lo, 1]),
        "ovality_abs": ov["ovality_abs"], "ovality_pct": ov["ovality_pct"],
        "k2_amp": ov["k2_amp"], "cost": 0.0, "boundary_points": cand.tolist(),
    }
    edge = {"z_center": z_center, "x": cand[:, 0], "y": cand[:, 1], "radius": r,
            "theta": theta, "inlier": fit["inlier_mask"].astype(int)}
    return result, edge, None


def analyze(points, z_step=2.0, z_min=None, z_max=None, min_points_per_slice=1000,
            max_points_for_speed=1_000_000, max_trials=4000):
    """Serial form of analyzer.py:181-230: optional closed z filter, windows from
    the configured bounds (both set) or the data, one process_slice per window,
    results in z order."""
    points = np.asarray(points)
    if points.ndim != 2 or points.shape[1] != 3:
        raise ValueError("Points must be Nx3 array")
    if z_min is not None and z_max is not None:
        points = points[(points[:, 2] >= z_min) & (points[:, 2] <= z_max)]
        if len(points) == 0:
            raise ValueError(f"No points in specified Z range [{z_min}, {z_max}]")
        lo_z, hi_z = z_min, z_max
    else:
        lo_z, hi_z = points[:, 2].min(), points[:, 2].max()
    centres, _, _ = slice_windows(lo_z, hi_z, z_step)
    out = []
    for zc in centres:
        res, _, warn = process_slice(points, zc, z_step, min_points_per_slice,
                                     max_points_for_speed, max_trials=max_trials)
        if warn is None:
            out.append(res)
    out.sort(key=lambda d: d["z_center"])
    return out


# --------------------------------------------------------------------------
# synthetic inputs shared by tests and bench
# --------------------------------------------------------------------------
def make_cylinder(n, slices, seed, radius=10.0, centre=(3.0, -2.0), sigma=0.05,
                  outlier_frac=0.0, ovality=0.0, z_step=2.0, pin=True):
    """Noisy cylinder scan: uniform z and angle, gaussian radial noise, a
    fraction of points replaced by uniform radii in [0, 1.4 R].  With ``pin`` the
    first two z values are set to the ends so exactly ``slices`` windows result."""
    gen = np.random.default_rng(seed)
    z = gen.uniform(0, slices * z_step, n)
    ang = gen.uniform(0, TWO_PI, n)
    r = radius * (1 + ovality * np.cos(2 * ang)) + gen.normal(0, sigma, n)
    bad = gen.random(n) < outlier_frac
    r[bad] = gen.uniform(0, 1.4 * radius, int(bad.sum()))
    if pin and n >= 2:
        z[0], z[1] = 0.0, slices * z_step
    return np.stack([centre[0] + r * np.cos(ang), centre[1] + r * np.sin(ang), z], axis=1)


def make_slices(slices, per_slice, seed, radius=10.0, centre=(3.0, -2.0), sigma=0.05,
                outlier_frac=0.0, jitter=0.0):
    """Pre-binned slices in CSR form: x, y (slices*per_slice,) and uniform offsets.
    ``jitter`` moves each slice's centre and radius a little so slices differ."""
    gen = np.random.default_rng(seed)
    total = slices * per_slice
    ang = gen.uniform(0, TWO_PI, total)
    r = radius + gen.normal(0, sigma, total)
    if outlier_frac > 0:
        bad = gen.random(total) < outlier_frac
        r[bad] = gen.uniform(0, 1.4 * radius, int(bad.sum()))
    sid = np.repeat(np.arange(slices), per_slice)
    shift = gen.normal(0, jitter, (slices, 3)) if jitter > 0 else np.zeros((slices, 3))
    r = r + shift[sid, 2]
    x = centre[0] + shift[sid, 0] + r * np.cos(ang)
    y = centre[1] + shift[sid, 1] + r * np.sin(ang)
    offsets = np.arange(slices + 1, dtype=np.int64) * per_slice
    return x, y, offsets


def year_boundary_rows(results, year, thickness):
    """Rows of year_{Y}_boundary_data.csv, built one by one as gui.py:1391-1436 does."""
    rows = []
    for result in results:
        cx, cy, R = result['cx'], result['cy'], result['R']
        if 'boundary_points' in result and result['boundary_points']:
            b = np.array(result['boundary_points'])
            dx, dy = b[:, 0] - cx, b[:, 1] - cy
            delta_r = np.hypot(dx, dy) - R
            theta = (np.arctan2(dy, dx) + 2 * np.pi) % (2 * np.pi)
            for i, (t, d) in enumerate(zip(theta, delta_r)):
                rows.append(dict(year=year, z_center=result['z_center'], cx=cx, cy=cy, R=R, thickness=thickness,
                                 theta=t, delta_r=d, boundary_point_index=i))
        else:
            rows.append(dict(year=year, z_center=result['z_center'], cx=cx, cy=cy, R=R, thickness=thickness,
                             theta=np.nan, delta_r=np.nan, boundary_point_index=0))
    return rows


def year_summary_rows(results, year, thickness):
    """Rows of year_{Y}_summary.csv, one per slice, as gui.py:1444-1455 builds them."""
    return [dict(year=year, z_center=r['z_center'], cx=r['cx'], cy=r['cy'], R=r['R'], thickness=thickness,
                 ovality_pct=r.get('ovality_pct', np.nan), n_boundary_points=len(r.get('boundary_points', [])))
            for r in results]


def year_match(df, current_z, current_thickness):
    """The comparison view's choice of rows from one year file (gui.py:1926-1945): None for a file without the
    required columns or without the slice, else (year as text, the slice's rows sorted by theta)."""
    required_cols = ['year', 'z_center', 'thickness', 'theta', 'delta_r']
    if not all(col in df.columns for col in required_cols):
        return None
    matching = df[(np.isclose(df['z_center'], current_z, atol=1e-3)) &
                  (np.isclose(df['thickness'], current_thickness, atol=1e-3))]
    if matching.empty:
        return None
    return str(matching['year'].iloc[0]), matching.sort_values('theta')
