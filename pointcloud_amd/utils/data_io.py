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

    import mmap

    with open(path, "rb") as fh:
        # the file's pages are parsed where the page cache holds them (no copy into a bytes object)
        data = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ) if os.fstat(fh.fileno()).st_size else b""
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


def _points_csv(points: list):
    """slice_points.csv as bytes, or None when the frames are not plain numeric columns (then pandas writes it).
    ``points``: DataFrames or dicts of equally long columns with the same keys."""
    from .. import _ffi

    names = list(points[0].keys())
    cols, is_int = [], []
    for name in names:
        parts = []
        for f in points:
            if list(f.keys()) != names:
                return None
            v = f[name]
            n = next(len(w) for w in f.values() if np.ndim(w) > 0) if isinstance(f, dict) else len(f)
            parts.append(np.full(n, v) if np.ndim(v) == 0 else np.asarray(v))
        col = np.concatenate(parts)
        if col.dtype.kind in "iub":
            if col.size and np.abs(col.astype(np.float64)).max() >= 2.0**53:
                return None
            is_int.append(1)
        elif col.dtype.kind == "f":
            is_int.append(0)
        else:
            return None
        cols.append(col.astype(np.float64))
    if "z_center" not in names or "theta" not in names:
        return None
    order = np.lexsort((cols[names.index("theta")], cols[names.index("z_center")]))     # stable, like sort_values
    body = _ffi.format_csv_rows([c[order] for c in cols], is_int)
    return (",".join(names) + "\n").encode() + body


def save_results(results: list, points: list, output_dir: str):
    """slice_results.csv (sorted by z_center) and slice_points.csv (sorted by
    z_center, theta), '%.6f' floats.  Reference: utils/data_io.py:29-47.  The
    points file - the big one - is formatted by the library on all host cores
    when its columns are plain numbers; ``points`` may hold DataFrames (as in the
    reference) or dicts of columns."""
    import pandas as pd

    if not results:
        return
    table = pd.DataFrame(results).sort_values("z_center").reset_index(drop=True)
    table.to_csv(os.path.join(output_dir, "slice_results.csv"), index=False, float_format="%.6f")
    if points:
        path = os.path.join(output_dir, "slice_points.csv")
        data = _points_csv(points)
        if data is not None:
            with open(path, "wb") as fh:
                fh.write(data)
            return
        frames = [f if isinstance(f, pd.DataFrame) else pd.DataFrame(f) for f in points]
        edge = pd.concat(frames, ignore_index=True).sort_values(["z_center", "theta"]).reset_index(drop=True)
        edge.to_csv(path, index=False, float_format="%.6f")
