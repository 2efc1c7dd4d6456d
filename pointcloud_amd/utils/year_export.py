"""Year-comparison export of residual profiles and the matcher that reads it back
(host code after the GPU path; the formats follow the reference's GUI:
gui.py:1388-1459 writes ``year_{Y}_boundary_data.csv`` and ``year_{Y}_summary.csv``,
gui.py:1921-1949 reads a boundary file back and picks the rows of one slice).

theta and delta_r of all boundary points of all slices come from ONE residual
call on the GPU (CSR batch of the results' boundary points); the tables are
assembled column-wise instead of row by row.
"""
from __future__ import annotations

import os

import numpy as np

from .. import _ffi

BOUNDARY_COLUMNS = ["year", "z_center", "cx", "cy", "R", "thickness", "theta", "delta_r", "boundary_point_index"]
SUMMARY_COLUMNS = ["year", "z_center", "cx", "cy", "R", "thickness", "ovality_pct", "n_boundary_points"]
REQUIRED_COLUMNS = ["year", "z_center", "thickness", "theta", "delta_r"]          # gui.py:1926


def year_boundary_table(results: list, year, thickness: float):
    """One row per boundary point: year, slice, fit, theta = (atan2 + 2 pi) % 2 pi and
    delta_r = hypot - R about the slice's own fit, and the point's index in the slice;
    a slice without boundary points gives one row with NaN theta / delta_r and index 0
    (gui.py:1391-1436)."""
    import pandas as pd

    S = len(results)
    counts = np.array([len(r["boundary_points"]) if r.get("boundary_points") else 0 for r in results], np.int64)
    offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    total = int(offsets[-1])
    fit = np.array([[r["cx"], r["cy"], r["R"]] for r in results], float).reshape(S, 3)
    theta = np.empty(0)
    dr = np.empty(0)
    if total:
        xy = np.concatenate([np.asarray(r["boundary_points"], float).reshape(-1, 2)
                             for r, c in zip(results, counts) if c])
        res = _ffi.residuals_batch(xy[:, 0], xy[:, 1], offsets, fit)
        theta, dr = res["theta"], res["dr"]
    rows = np.maximum(counts, 1)                       # an empty slice still gets its one row
    row_off = np.concatenate([[0], np.cumsum(rows)])
    n_rows = int(row_off[-1])
    slice_of = np.repeat(np.arange(S), rows)
    th_col = np.full(n_rows, np.nan)
    dr_col = np.full(n_rows, np.nan)
    idx_col = np.zeros(n_rows, np.int64)
    if total:
        has = np.repeat(counts > 0, rows)
        th_col[has] = theta
        dr_col[has] = dr
        idx_col[has] = np.arange(total) - np.repeat(offsets[:-1], counts)
    z = np.array([r["z_center"] for r in results], float)
    return pd.DataFrame({
        "year": np.full(n_rows, year, dtype=object) if isinstance(year, str) else np.full(n_rows, year),
        "z_center": z[slice_of] if S else np.empty(0),
        "cx": fit[slice_of, 0] if S else np.empty(0),
        "cy": fit[slice_of, 1] if S else np.empty(0),
        "R": fit[slice_of, 2] if S else np.empty(0),
        "thickness": np.full(n_rows, thickness),
        "theta": th_col,
        "delta_r": dr_col,
        "boundary_point_index": idx_col,
    }, columns=BOUNDARY_COLUMNS)


def year_summary_table(results: list, year, thickness: float):
    """One row per slice (gui.py:1444-1455)."""
    import pandas as pd

    return pd.DataFrame([{
        "year": year,
        "z_center": r["z_center"],
        "cx": r["cx"],
        "cy": r["cy"],
        "R": r["R"],
        "thickness": thickness,
        "ovality_pct": r.get("ovality_pct", np.nan),
        "n_boundary_points": len(r.get("boundary_points", [])),
    } for r in results], columns=SUMMARY_COLUMNS)


def export_year_data(results: list, year, thickness: float, save_dir: str):
    """Write both files with '%.6f' floats (gui.py:1439-1459); returns their paths."""
    year = str(year).strip()
    boundary_path = os.path.join(save_dir, f"year_{year}_boundary_data.csv")
    summary_path = os.path.join(save_dir, f"year_{year}_summary.csv")
    year_boundary_table(results, year, thickness).to_csv(boundary_path, index=False, float_format="%.6f")
    year_summary_table(results, year, thickness).to_csv(summary_path, index=False, float_format="%.6f")
    return boundary_path, summary_path


def match_year_slices(table, z_center: float, thickness: float, atol: float = 1e-3):
    """Rows of a boundary table that belong to the slice (z_center, thickness), sorted by
    theta, and the year they carry; None if the table lacks a required column or holds no
    such slice (gui.py:1926-1945: ``isclose(atol=1e-3)`` on both keys)."""
    if isinstance(table, (str, os.PathLike)):
        import pandas as pd
        table = pd.read_csv(table)
    if not all(c in table.columns for c in REQUIRED_COLUMNS):
        return None
    hit = table[np.isclose(table["z_center"], z_center, atol=atol) & np.isclose(table["thickness"], thickness, atol=atol)]
    if hit.empty:
        return None
    return str(hit["year"].iloc[0]), hit.sort_values("theta")
