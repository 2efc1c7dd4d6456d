"""Boundary extraction and ovality analysis on the GPU behind the reference's
function signatures (reference: utils/boundary.py)."""
from __future__ import annotations

import numpy as np

from .. import _ffi


def boundary_by_angle_max(points_xy: np.ndarray, center=None, bins: int = 720) -> np.ndarray:
    """Reference: utils/boundary.py:7-42.  Returns the point of largest radius of
    every non-empty angular bin, in order of angle."""
    XY = np.asarray(points_xy, float)
    if center is None:
        if XY.shape[0] < 3:
            raise ValueError("Need at least 3 points for Pratt fit")
        fit, _ = _ffi.kasa_batch(XY[:, 0], XY[:, 1])
        cx, cy = float(fit[0, 0]), float(fit[0, 1])
    else:
        cx, cy = center
    sel = _ffi.angle_max_batch(XY[:, 0], XY[:, 1], None, [cx, cy, 0.0], bins)[0]
    sel = sel[sel >= 0]              # bins ascend with the angle, so this is the reference's argsort order
    if sel.size == 0:
        raise RuntimeError("No boundary points extracted (data too sparse?)")
    return XY[sel]


def boundary_by_convex_hull(points_xy: np.ndarray) -> np.ndarray:
    """Reference: utils/boundary.py:44-54 (OpenCV hull of the float32-rounded
    points).  Returns the hull vertices counter-clockwise from the
    lexicographically smallest one; OpenCV's starting vertex is not reproduced."""
    p32 = np.asarray(points_xy).astype(np.float32).reshape(-1, 2).astype(np.float64)
    h = _ffi.unique_hull_batch(p32[:, 0], p32[:, 1])
    nh = int(h["n_hull"][0])
    idx = h["hull_idx"][:nh]
    edge_xy = np.stack([h["ux"][idx], h["uy"][idx]], axis=1)
    if len(edge_xy) < 10:
        raise RuntimeError("Too few hull vertices")
    return edge_xy


def ovality_and_fourier(x: np.ndarray, y: np.ndarray,
                        center_x: float, center_y: float,
                        radius: float) -> dict:
    """Reference: utils/boundary.py:56-90."""
    x = np.asarray(x, float).ravel()
    y = np.asarray(y, float).ravel()
    scal, dev = _ffi.ovality_batch(x, y, None, [center_x, center_y, radius])
    K = max(1024, x.size)
    return {
        "Rmax": np.float64(scal[0, 0]), "Rmin": np.float64(scal[0, 1]),
        "ovality_abs": np.float64(scal[0, 2]),
        "ovality_pct": np.float64(scal[0, 3]),
        "k2_amp": np.float64(scal[0, 4]),
        "theta": np.linspace(0, 2 * np.pi, K, endpoint=False),
        "dev_uniform": dev[0, :K],
    }
