"""Circle fitting on the GPU behind the reference's function signatures
(reference: utils/circle_fitting.py).  Every function forwards its arrays to
libpcfit_hip.so; nothing is computed on the host except argument checks and
the packaging of results.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np

from .. import _ffi


def _has_enough_2d_variation(xy: np.ndarray, eps: float = 1e-12) -> bool:
    """Reference: utils/circle_fitting.py:14-19."""
    xy = np.asarray(xy, float)
    if xy.shape[0] < 3:
        return False
    return bool(_ffi.rank_check_batch(xy[:, 0], xy[:, 1], None, eps)[0])


def _circle_from_3pts(p1, p2, p3):
    """Reference: utils/circle_fitting.py:22-32."""
    tri = np.array([p1[0], p1[1], p2[0], p2[1], p3[0], p3[1]], float)
    out, ok = _ffi.circle3_batch(tri)
    if not ok[0]:
        return None
    return float(out[0, 0]), float(out[0, 1]), float(out[0, 2])


def fit_circle_pratt(x: np.ndarray, y: np.ndarray) -> Tuple[float, float, float]:
    """Reference: utils/circle_fitting.py:35-54 (an algebraic Kasa fit despite the name)."""
    x = np.asarray(x, float).ravel()
    y = np.asarray(y, float).ravel()
    if x.size < 3:
        raise ValueError("Need at least 3 points for Pratt fit")
    out, _ = _ffi.kasa_batch(x, y)
    return float(out[0, 0]), float(out[0, 1]), float(out[0, 2])


def lm_circle_geometric(
    x: np.ndarray, y: np.ndarray,
    cx0: float, cy0: float, r0: float,
    huber_delta: float = 1.0,
    max_iter: int = 100,
    lm_lambda0: float = 1e-2,
    tol_update: float = 1e-10,
    tol_cost: float = 1e-12,
) -> Tuple[float, float, float]:
    """Reference: utils/circle_fitting.py:57-109."""
    x = np.asarray(x, float).ravel()
    y = np.asarray(y, float).ravel()
    out, _ = _ffi.lm_batch(x, y, None, [float(cx0), float(cy0), float(r0)], huber_delta, max_iter, lm_lambda0,
                           tol_update, tol_cost)
    return float(out[0, 0]), float(out[0, 1]), float(out[0, 2])


def _ransac_circle(
    xy: np.ndarray,
    inlier_tol: float = 0.5,
    max_trials: int = 4000,
    min_inliers_frac: float = 0.1,
    r_range: Optional[Tuple[float, float]] = None,
    rng_seed: int = 42,
) -> Optional[Dict[str, np.ndarray]]:
    """Reference: utils/circle_fitting.py:112-143."""
    xy = np.asarray(xy, float)
    if xy.shape[0] < 3:
        return None
    r = _ffi.ransac_batch(xy[:, 0], xy[:, 1], None, inlier_tol, max_trials, min_inliers_frac, r_range, rng_seed)
    if r["status"][0] != 0:
        return None
    m = r["model"][0]
    return {"score": int(r["score"][0]), "cx": float(m[0]), "cy": float(m[1]), "R": float(m[2]), "mask": r["mask"]}


def fit_circle_hybrid_at_z(
    points_xyz: np.ndarray,
    z_elevation: float,
    z_tolerance: float,
    inlier_tol: float = 0.5,
    max_trials: int = 4000,
    min_inliers_frac: float = 0.1,
    r_range: Optional[Tuple[float, float]] = None,
    huber_delta: float = 1.0,
    use_hull_vertices_only: bool = True,
    *,
    hull_start="qhull",
) -> Optional[Dict[str, np.ndarray]]:
    """Reference: utils/circle_fitting.py:146-232.

    ``hull_start`` (not in the reference): where the candidate walk around the
    hull begins.  "qhull" (default): where ``scipy.spatial.ConvexHull(xy).vertices``
    begins (circle_fitting.py:189-190), found by replaying Qhull's construction -
    the candidates, the sampled triples and with them the fit are then the
    reference's own.  "canonical": at the lexicographically smallest vertex (no
    replay).  An int: at that vertex, an index into the sorted distinct slice
    points.
    """
    points_xyz = np.asarray(points_xyz)
    if points_xyz.ndim != 2 or points_xyz.shape[1] != 3:
        raise ValueError("points_xyz must be (N,3)")
    lo = np.array([z_elevation - z_tolerance], float)
    hi = np.array([z_elevation + z_tolerance], float)
    offsets, x, y, _ = _ffi.bin_slices(points_xyz, lo, hi, want_perm=False)
    if offsets[1] < 3:
        return None
    if isinstance(hull_start, str):
        hs, order = None, hull_start                       # "qhull" / "canonical" (anything else is refused below)
    else:
        if hull_start is None:
            raise ValueError('hull_start: "qhull", "canonical" or a vertex index')
        hs, order = np.array([hull_start], np.int64), "canonical"
    h = _ffi.hybrid_batch(x, y, offsets, inlier_tol, max_trials, min_inliers_frac, r_range, huber_delta,
                          use_hull_vertices_only, hs, want_profile=False, order=order)
    if h["status"][0] != 0:
        return None
    nu = int(h["n_unique"][0])
    return {
        "center_x": float(h["fit"][0, 0]),
        "center_y": float(h["fit"][0, 1]),
        "radius": float(h["fit"][0, 2]),
        "inlier_mask": h["mask"],
        "fit_points_xy": np.stack([h["ux"][:nu], h["uy"][:nu]], axis=1),
        "candidates_xy": np.stack([h["cand_x"], h["cand_y"]], axis=1),
        "z_elevation": float(z_elevation),
    }


def visualize_hybrid_fit(res):
    """Reference: utils/circle_fitting.py:235-264 draws a matplotlib figure.
    Plotting is not part of this package."""
    raise NotImplementedError("plotting is outside the scope of pointcloud_amd; use the dict returned by "
                              "fit_circle_hybrid_at_z with a plotting tool of your choice")
