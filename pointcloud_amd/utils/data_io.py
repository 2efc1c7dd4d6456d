"""Text loading and CSV export around the GPU path (host code; the formats
follow the reference: utils/data_io.py:8-47).  These are the steps before and
after the hot path; they stay on the host."""
from __future__ import annotations

import os

import numpy as np


def _parse_line(raw: str):
    """One line by the reference's rule (utils/data_io.py:13-24): the point, or None."""
    text = raw.strip()
    if not text or text.startswith(("#", "//", ";")):
        return None
    fields = text.replace(",", " ").split()
    if len(fields) < 3:
        return None
    try:
        return float(fields[0]), float(fields[1]), float(fields[2])
    except Exception:
        return None


def load_txt_points(path: str, n_threads: int = 0) -> np.ndarray:
    """Tolerant XYZ reader.  Reference: utils/data_io.py:8-27 - blank lines and
    lines starting with '#', '//' or ';' are skipped, commas count as blanks,
    the first three fields must parse as floats, anything else is ignored.
    Plain numeric lines are parsed by the library on all host cores; lines with
    any other content come back flagged and get the literal rule here."""
    from .. import _ffi

    with open(path, "rb") as fh:
        data = fh.read()
    rows, deferred = _ffi.parse_xyz_text(data, n_threads)
    idx = np.flatnonzero(deferred)
    if idx.size:
        keep = np.ones(len(rows), bool)
        for i in idx:
            line = data[int(rows[i, 1]):int(rows[i, 2])].decode("utf-8", errors="ignore")
            pt = _parse_line(line)
            if pt is None:
                keep[i] = False
            else:
                rows[i] = pt
        rows = rows[keep]
    if len(rows) == 0:
        raise RuntimeError("No valid XYZ lines found in TXT.")
    return np.ascontiguousarray(rows, dtype=np.float64)


def save_results(results: list, points: list, output_dir: str):
    """slice_results.csv (sorted by z_center) and slice_points.csv (sorted by
    z_center, theta), '%.6f' floats.  Reference: utils/data_io.py:29-47."""
    import pandas as pd

    if not results:
        return
    table = pd.DataFrame(results).sort_values("z_center").reset_index(drop=True)
    table.to_csv(os.path.join(output_dir, "slice_results.csv"), index=False, float_format="%.6f")
    if points:
        edge = pd.concat(points, ignore_index=True).sort_values(["z_center", "theta"]).reset_index(drop=True)
        edge.to_csv(os.path.join(output_dir, "slice_points.csv"), index=False, float_format="%.6f")
