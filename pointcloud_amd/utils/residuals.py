"""Residual profiles delta_r(theta) = r - R and their histograms.

The reference keeps this arithmetic inside its GUI (gui.py:972-987 for the
boundary points of a slice result, gui.py:1398-1423 for the export rows) and in
a scratch script (1.py:301-345, which can also use every slice point).  The
per-point work runs on the GPU; ordering by theta is a host argsort.
"""
from __future__ import annotations

import numpy as np

from .. import _ffi


def residual_profile(points_xy, cx, cy, R, sort=True):
    """theta in [0, 2pi) and delta_r for every point; ordered by theta when ``sort``."""
    xy = np.asarray(points_xy, float).reshape(-1, 2)
    r = _ffi.residuals_batch(xy[:, 0], xy[:, 1], None, [cx, cy, R])
    theta, dr = r["theta"], r["dr"]
    if sort:
        order = np.argsort(theta)
        return theta[order], dr[order]
    return theta, dr


def compute_residual_profile(result):
    """Reference: gui.py:972-987 (method of the GUI class; takes a slice result dict)."""
    if "boundary_points" not in result or not result["boundary_points"]:
        return None
    return residual_profile(np.array(result["boundary_points"]), result["cx"], result["cy"], result["R"])


def filtered_residual_profile(points_xy, cx, cy, R, inlier_mask=None, quantile=0.95):
    """The profile the slice view draws (gui.py:1626-1655, data part): delta_r of the edge points, kept where
    ``inlier_mask`` says so when it has one entry per point, otherwise where |delta_r| is at most its
    ``quantile`` quantile; ordered by theta, theta in degrees.  Returns (theta_deg, delta_r, kept) with
    ``kept`` the boolean mask over the input points."""
    xy = np.asarray(points_xy, float).reshape(-1, 2)
    if len(xy) == 0:
        return np.zeros(0), np.zeros(0), np.zeros(0, bool)
    theta, dr = residual_profile(xy, cx, cy, R, sort=False)
    if inlier_mask is not None and len(inlier_mask) == len(xy):
        kept = np.asarray(inlier_mask, bool)
    else:
        mag = np.abs(dr)
        kept = mag <= np.quantile(mag, quantile)
    theta, dr = theta[kept], dr[kept]
    order = np.argsort(theta)
    return np.degrees(theta[order]), dr[order], kept


def residual_histogram(points_xy, cx, cy, R, bins, lo, hi):
    """numpy.histogram(delta_r, bins=bins, range=(lo, hi)) computed on the GPU."""
    xy = np.asarray(points_xy, float).reshape(-1, 2)
    r = _ffi.residuals_batch(xy[:, 0], xy[:, 1], None, [cx, cy, R], bins, lo, hi, want_theta=False, want_dr=False)
    return r["hist"][0]


def residual_profile_at_z(points_xyz, res_fit, z_elevation, z_tolerance, use_hull_vertices=True):
    """Reference: 1.py:301-345."""
    cx = float(res_fit["center_x"])
    cy = float(res_fit["center_y"])
    R = float(res_fit["radius"])
    pts = np.asarray(points_xyz)
    lo = np.array([z_elevation - z_tolerance], float)
    hi = np.array([z_elevation + z_tolerance], float)
    offsets, x, y, _ = _ffi.bin_slices(pts, lo, hi, want_perm=False)
    if offsets[1] < 3:
        return None
    h = _ffi.unique_hull_batch(x, y)
    nu, nh = int(h["n_unique"][0]), int(h["n_hull"][0])
    used = np.stack([h["ux"][:nu], h["uy"][:nu]], axis=1)
    if use_hull_vertices and h["status"][0] == 0 and nh >= 3:
        used = used[h["hull_idx"][:nh]]
    theta, dr = residual_profile(used, cx, cy, R, sort=False)
    order = np.argsort(theta)
    return {"theta": theta[order], "delta_r": dr[order], "xy_used": used[order], "cx": cx, "cy": cy, "R": R}
