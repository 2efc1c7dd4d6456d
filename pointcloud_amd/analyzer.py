"""Cylinder analysis driver with the reference's interface (reference:
analyzer.py).  ``Config`` and ``CylinderAnalyzer`` keep their fields, method
names, argument meaning, return types and error messages; underneath, all
slices of a scan go through the GPU in one batch: stable slice binning, then the
hybrid fit and the per-slice metrics (see include/pcfit.h).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

from . import _ffi
from .utils.data_io import load_txt_points, save_results


@dataclass
class Config:
    """Analysis configuration parameters (reference: analyzer.py:17-30).  Only
    z_step, min_points_per_slice, max_points_for_speed, z_min and z_max influence
    the numbers; the other fields are accepted for compatibility."""
    z_step: float = 2.0
    max_points_for_speed: int = 1_000_000
    min_points_per_slice: int = 1000
    inlier_quantile: float = 0.80
    boundary_method: str = 'angle_max'
    angle_bins: int = 720
    draw_per_slice_images: bool = True
    slice_plot_sample_points: int = 30000
    overlay_all: bool = True
    z_min: Optional[float] = None
    z_max: Optional[float] = None
    # not in the reference: "qhull" lists every slice's hull vertices from the vertex scipy's ConvexHull starts
    # with (analyzer.py:67 -> circle_fitting.py:189-192), so candidates, sampled triples and fits are the
    # reference's own; "canonical" starts at the lexicographically smallest vertex (no replay of Qhull, and
    # slices longer than 1024 points drop their interior points before they are sorted: faster)
    hull_order: str = "qhull"


# fit parameters the reference hard-codes in process_slice (analyzer.py:67-71)
_FIT = dict(inlier_tol=0.5, max_trials=4000, min_inliers_frac=0.1, huber_delta=1.0, use_hull=True)


class CylinderAnalyzer:
    def __init__(self, config: Config = None):
        self.config = config or Config()
        self.z_min = self.config.z_min
        self.z_max = self.config.z_max
        # output names of the reference (analyzer.py:38-41); nothing is created on disk here
        self.slice_fig_dir = 'plots_slices'
        self.results_csv = 'slice_results.csv'
        self.points_csv = 'slice_points.csv'
        self.overlay_png = 'all_circles_overlay.png'

    def _slice_result(self, s: int, z_center, n0: int, h: dict, want_ov: bool = True):
        """Result dict, edge frame columns and the ovality dict of slice ``s``
        (reference: analyzer.py:77-127).  ``want_ov=False`` leaves out the two
        profile arrays of the ovality dict (the batch drivers do not return it)."""
        dz = self.config.z_step / 2.0
        c0, c1 = int(h["cand_offsets"][s]), int(h["cand_offsets"][s + 1])
        cand = np.stack([h["cand_x"][c0:c1], h["cand_y"][c0:c1]], axis=1)
        cx, cy, R = (float(v) for v in h["fit"][s])
        i_max, i_min = int(h["argmax"][s]), int(h["argmin"][s])
        ovs = h["ov"][s]
        K = max(1024, c1 - c0)
        ov = {
            "Rmax": np.float64(ovs[0]), "Rmin": np.float64(ovs[1]),
            "ovality_abs": np.float64(ovs[2]), "ovality_pct": np.float64(ovs[3]), "k2_amp": np.float64(ovs[4]),
        }
        if want_ov:
            ov["theta"] = np.linspace(0, 2 * np.pi, K, endpoint=False)
            ov["dev_uniform"] = None if h["dev_uniform"] is None else h["dev_uniform"][s, :K]
        result = {
            "z_center": z_center,
            "z_low": z_center - dz,
            "z_high": z_center + dz,
            "n_points_slice": int(n0),
            "n_edge": int(len(cand)),
            "cx": cx, "cy": cy, "R": R,
            "Rmax": ov["Rmax"],
            "Rmin": ov["Rmin"],
            "x_at_Rmax": float(cand[i_max, 0]),
            "y_at_Rmax": float(cand[i_max, 1]),
            "x_at_Rmin": float(cand[i_min, 0]),
            "y_at_Rmin": float(cand[i_min, 1]),
            "ovality_abs": ov["ovality_abs"],
            "ovality_pct": ov["ovality_pct"],
            "k2_amp": ov["k2_amp"],
            "cost": 0.0,
            "boundary_points": cand.tolist(),
        }
        edge = {
            "z_center": z_center,
            "x": cand[:, 0],
            "y": cand[:, 1],
            "radius": h["radius"][c0:c1],
            "theta": h["theta"][c0:c1],
            "inlier": h["mask"][c0:c1].astype(int),
        }
        return result, cand, (cx, cy, R), ov, edge

    # ------------------------------------------------------------------ reference API
    def process_slice(self, points: np.ndarray, z_center: float, *, hull_start=None
                      ) -> Tuple[dict, "pd.DataFrame", str]:  # noqa: F821
        """Process a single slice of points (reference: analyzer.py:46-129).  ``hull_start``: an index, "qhull"
        or "canonical", as in utils.circle_fitting.fit_circle_hybrid_at_z; None: what the configuration says."""
        import pandas as pd

        if not isinstance(points, _ffi.DeviceScan):
            points = np.asarray(points, float)
        if hull_start is None:
            hull_start = self.config.hull_order
        if isinstance(hull_start, str):
            hs, order = None, hull_start
        else:
            hs, order = np.array([hull_start], np.int64), "canonical"
        dz = self.config.z_step / 2.0
        offsets, x, y, _ = _ffi.bin_slices(points, np.array([z_center - dz]), np.array([z_center + dz]),
                                           want_perm=False)
        n0 = int(offsets[1])
        if n0 < self.config.min_points_per_slice:
            return None, None, f"SKIP zc={z_center:.3f}: too sparse ({n0} points)"
        h = _ffi.hybrid_batch(x, y, offsets, _FIT["inlier_tol"], _FIT["max_trials"], _FIT["min_inliers_frac"], None,
                              _FIT["huber_delta"], _FIT["use_hull"], hs, want_profile=True, want_unique=False,
                              order=order)
        slc = np.stack([x, y], axis=1)
        if n0 > self.config.max_points_for_speed:
            # the reference thins only what it hands back for plotting; the fit sees every point
            idx = np.random.choice(n0, self.config.max_points_for_speed, replace=False)
            slc = slc[idx]
            n0 = len(slc)
        if h["status"][0] != 0:
            return None, None, f"SKIP zc={z_center:.3f}: hybrid fit failed"
        result, cand, fit, ov, edge = self._slice_result(0, z_center, n0, h)
        return (result, cand, slc, fit, ov), pd.DataFrame(edge), None

    def analyze_file(self, file_path: str):
        """Process entire point cloud file (reference: analyzer.py:131-179).  Writes
        slice_results.csv and slice_points.csv into the working directory; the
        reference's matplotlib figures are not produced."""
        points = load_txt_points(file_path)
        # the scan goes to the GPU once; its z range comes back from there
        points = _ffi.DeviceScan(points)
        z_min, z_max = np.float64(points.z_min), np.float64(points.z_max)
        print(f"Z range in data: [{z_min:.3f}, {z_max:.3f}]  "
              f"| step={self.config.z_step}")
        dz = self.config.z_step / 2.0
        z_centers = np.arange(z_min + dz, z_max - dz + 1e-9, self.config.z_step)
        results, frames = [], []
        for res, edge, warn in self._analyze_windows(points, z_centers, want_edges=True):
            if warn is not None:
                print(warn)
                continue
            results.append(res)
            frames.append(edge)                       # (columns of one slice; save_results takes them as they are)
        if results:
            save_results(results, frames, ".")
        return results

    def _hybrid_windows(self, points, z_centers):
        """The hybrid fit and the metrics of every window of a scan (an (N,3) array or a DeviceScan), as the dict
        of arrays _ffi.scan_hybrid_batch returns: the scan is uploaded once (if it is not resident already) and
        binned on the device; only per-window results come back."""
        dz = self.config.z_step / 2.0
        z_centers = np.asarray(z_centers, float)
        return _ffi.scan_hybrid_batch(points, z_centers - dz, z_centers + dz, _FIT["inlier_tol"], _FIT["max_trials"],
                                      _FIT["min_inliers_frac"], None, _FIT["huber_delta"], _FIT["use_hull"],
                                      order=self.config.hull_order)

    def _results_of(self, z_centers, h, want_edges=False):
        """(result, edge columns or None, warning) per window, in window order, from the arrays of _hybrid_windows."""
        # Building hundreds of thousands of small lists and dicts makes the cyclic garbage collector walk them
        # again and again (a 4x slowdown measured on the candidate lists alone); it is paused while the
        # results are put together, and they are handed out afterwards.
        import gc

        was_enabled = gc.isenabled()
        gc.disable()
        try:
            return list(self._window_results(np.asarray(z_centers, float), h, want_edges))
        finally:
            if was_enabled:
                gc.enable()

    def _analyze_windows(self, points, z_centers, want_edges=False):
        """Yield (result, edge columns or None, warning) per window, in window order."""
        if len(z_centers) == 0:
            return
        yield from self._results_of(z_centers, self._hybrid_windows(points, z_centers), want_edges)

    def _window_results(self, z_centers, h, want_edges):
        dz = self.config.z_step / 2.0
        counts = np.diff(h["offsets"])
        # the columns of all result dicts at once (reference: analyzer.py:77-127 builds them slice by slice);
        # per window only the dict itself is put together
        co = h["cand_offsets"]
        good = np.flatnonzero((counts >= self.config.min_points_per_slice) & (h["status"] == 0))
        at = {int(s): j for j, s in enumerate(good)}
        cand_x, cand_y = h["cand_x"], h["cand_y"]
        i_max, i_min = co[good] + h["argmax"][good], co[good] + h["argmin"][good]
        x_max, y_max = cand_x[i_max].tolist(), cand_y[i_max].tolist()
        x_min, y_min = cand_x[i_min].tolist(), cand_y[i_min].tolist()
        fit = h["fit"].tolist()
        ov = [list(h["ov"][:, k]) for k in range(5)]
        pairs = np.stack([cand_x, cand_y], axis=1).tolist()
        z_low, z_high = list(z_centers - dz), list(z_centers + dz)
        n_kept = np.minimum(counts, self.config.max_points_for_speed).tolist()
        co = co.tolist()
        for s, zc in enumerate(z_centers):
            j = at.get(s)
            if j is None:
                n0 = int(counts[s])
                if n0 < self.config.min_points_per_slice:
                    yield None, None, f"SKIP zc={zc:.3f}: too sparse ({n0} points)"
                else:
                    yield None, None, f"SKIP zc={zc:.3f}: hybrid fit failed"
                continue
            c0, c1 = co[s], co[s + 1]
            cx, cy, R = fit[s]
            result = {
                "z_center": zc,
                "z_low": z_low[s],
                "z_high": z_high[s],
                "n_points_slice": n_kept[s],
                "n_edge": c1 - c0,
                "cx": cx, "cy": cy, "R": R,
                "Rmax": ov[0][s],
                "Rmin": ov[1][s],
                "x_at_Rmax": x_max[j],
                "y_at_Rmax": y_max[j],
                "x_at_Rmin": x_min[j],
                "y_at_Rmin": y_min[j],
                "ovality_abs": ov[2][s],
                "ovality_pct": ov[3][s],
                "k2_amp": ov[4][s],
                "cost": 0.0,
                "boundary_points": pairs[c0:c1],
            }
            edge = None
            if want_edges:
                edge = {
                    "z_center": zc,
                    "x": cand_x[c0:c1],
                    "y": cand_y[c0:c1],
                    "radius": h["radius"][c0:c1],
                    "theta": h["theta"][c0:c1],
                    "inlier": h["mask"][c0:c1].astype(int),
                }
            yield result, edge, None

    def _resident(self, points):
        """The scan on the GPU, as analyze_with_progress takes it (reference: analyzer.py:185-196): an (N,3) array
        goes up once - with Config.z_min / z_max only its rows inside that closed range, in their order, picked
        by the library's host threads while they stage the upload (no masked copy of the scan on the host) - and
        a DeviceScan passes through."""
        if isinstance(points, _ffi.DeviceScan):
            # a scan already on the GPU; a configured z range would have to filter it, which needs the array
            if self.z_min is not None and self.z_max is not None:
                raise ValueError("a DeviceScan cannot be combined with Config.z_min / z_max: pass the array")
            return points
        if not isinstance(points, np.ndarray):
            points = np.array(points)

        if points.ndim != 2 or points.shape[1] != 3:
            raise ValueError("Points must be Nx3 array")

        if self.z_min is not None and self.z_max is not None:
            points = _ffi.DeviceScan(points, z_range=(self.z_min, self.z_max))
            if len(points) == 0:
                raise ValueError(f"No points in specified Z range [{self.z_min}, {self.z_max}]")
        elif len(points) > 0:
            # upload once; the z range comes back from the device instead of two strided host passes
            points = _ffi.DeviceScan(points)
        return points

    def analyze_with_progress(self, points: np.ndarray, progress_callback=None):
        """Analyze points with progress reporting (reference: analyzer.py:181-230).
        All windows run in one GPU batch; the callback then receives the same
        integer percentages the reference's thread pool would emit."""
        points = self._resident(points)

        results = []
        z_centers = self.calculate_z_centers(points)
        total = len(z_centers)
        done = 0
        for res, _, warn in self._analyze_windows(points, z_centers):
            if warn is None:
                results.append(res)
            done += 1
            if progress_callback:
                progress_callback(int(done / total * 100))

        results.sort(key=lambda r: r['z_center'])
        return results

    def calculate_z_centers(self, points: np.ndarray) -> np.ndarray:
        """Calculate z-centers for slicing (reference: analyzer.py:232-243)."""
        if self.z_min is not None and self.z_max is not None:
            z_min, z_max = self.z_min, self.z_max
        elif isinstance(points, _ffi.DeviceScan):
            z_min, z_max = np.float64(points.z_min), np.float64(points.z_max)
        else:
            z_min, z_max = points[:, 2].min(), points[:, 2].max()

        dz = self.config.z_step / 2.0
        return np.arange(z_min + dz,
                         z_max - dz + 1e-9,
                         self.config.z_step)
