#!/usr/bin/env python3
"""Record outputs of the real reference as golden vectors (build container only).

Run:  MPLBACKEND=Agg python tests/golden/gen_golden.py [/root/reference]

Imports the reference's ``utils.circle_fitting``, ``utils.boundary`` and
``analyzer`` (with an empty stand-in module for the absent ``cv2``, which only
``boundary_by_convex_hull`` touches), runs them on seeded synthetic inputs and
writes ``tests/golden/*.npz``: inputs (or the seed + a SHA-256 of the input
bytes when the input is large) and the reference's outputs.  Nothing else in
the repository reads the reference.
"""
from __future__ import annotations

import hashlib
import os
import sys
import tempfile
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"

warnings.simplefilter("ignore")
os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
sys.path.insert(0, REF)
sys.path.insert(1, ROOT)
os.chdir(tempfile.mkdtemp(prefix="pcfit_golden_"))      # the analyzer creates ./plots_slices

from utils import circle_fitting as rcf      # noqa: E402  (reference)
from utils import boundary as rbd            # noqa: E402  (reference)
import analyzer as ran                       # noqa: E402  (reference)
import scipy                                 # noqa: E402
from oracle.ref_np import make_cylinder, make_slices   # noqa: E402  (input generators only)

META = dict(numpy=np.__version__, scipy=scipy.__version__, python=sys.version.split()[0])


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, meta=np.array(repr(META)), **arrays)
    print(f"{name}.npz  {os.path.getsize(path)/1024:.1f} KiB")


def ring(n, seed, r=10.0, c=(3.0, -2.0), sigma=0.05, arc=(0.0, 2 * np.pi), outlier=0.0):
    g = np.random.default_rng(seed)
    th = g.uniform(arc[0], arc[1], n)
    rr = r + g.normal(0, sigma, n)
    if outlier > 0:
        bad = g.random(n) < outlier
        rr[bad] = g.uniform(0, 1.4 * r, int(bad.sum()))
    return np.stack([c[0] + rr * np.cos(th), c[1] + rr * np.sin(th)], axis=1)


# 1. trial -> triple tables ------------------------------------------------
def g_choice3():
    out = {}
    for n in (3, 4, 5, 7, 64, 86, 100, 257, 1000, 8192, 10001, 65537, 1000003):
        g = np.random.default_rng(42)
        out[f"n{n}"] = np.array([g.choice(n, size=3, replace=False) for _ in range(64)], dtype=np.int32)
    g = np.random.default_rng(42)
    out["n92_t4000"] = np.array([g.choice(92, size=3, replace=False) for _ in range(4000)], dtype=np.int32)
    for seed in (0, 1, 7, 123456789, 2**40 + 5):
        g = np.random.default_rng(seed)
        out[f"seed{seed}_n500"] = np.array([g.choice(500, size=3, replace=False) for _ in range(32)], dtype=np.int32)
    save("choice3", **out)


# 2. three-point circles ---------------------------------------------------
def g_circle3():
    g = np.random.default_rng(11)
    tri = g.normal(0, 10, (400, 3, 2))
    tri[300:350, 2] = tri[300:350, 0] + (tri[300:350, 1] - tri[300:350, 0]) * g.uniform(0, 2, (50, 1)) \
        + g.normal(0, 1e-9, (50, 2))                              # nearly collinear
    tri[350:380, 2] = tri[350:380, 0] + (tri[350:380, 1] - tri[350:380, 0]) * 0.5   # collinear up to rounding
    tri[380:] *= 1e-7                                             # tiny coordinates -> |det| < 1e-12
    ok = np.zeros(400, bool)
    val = np.full((400, 3), np.nan)
    for i, t in enumerate(tri):
        c = rcf._circle_from_3pts(t[0], t[1], t[2])
        if c is not None:
            ok[i] = True
            val[i] = c
    save("circle3", tri=tri, ok=ok, val=val)


# 3 + 4. Kasa and LM ---------------------------------------------------------
def fit_cases():
    cases = [
        ring(8192, 1), ring(8192, 2, outlier=0.3), ring(500, 3, arc=(0.3, 0.3 + np.pi / 2)),
        ring(92, 4), ring(64, 5, sigma=0.5), ring(3, 6), ring(4000, 7, r=2.5, c=(-100.0, 250.0), sigma=0.01),
        ring(1000, 8, r=1e3, c=(1e5, -2e5), sigma=1.0), ring(257, 9, outlier=0.1),
        ring(5000, 10, arc=(0, np.pi / 8), sigma=0.02),
        np.stack([np.linspace(0, 1, 50), 2 * np.linspace(0, 1, 50) + 1], axis=1) + np.random.default_rng(12).normal(0, 1e-3, (50, 2)),
        np.concatenate([ring(40, 13, sigma=0.0)] * 3) + np.random.default_rng(14).normal(0, 1e-6, (120, 2)),
    ]
    return cases


def g_kasa_lm():
    out = {}
    for i, xy in enumerate(fit_cases()):
        out[f"xy{i}"] = xy
        k = rcf.fit_circle_pratt(xy[:, 0], xy[:, 1])
        out[f"kasa{i}"] = np.array(k)
        out[f"lm{i}"] = np.array(rcf.lm_circle_geometric(xy[:, 0], xy[:, 1], *k))
        out[f"lm_d02_{i}"] = np.array(rcf.lm_circle_geometric(xy[:, 0], xy[:, 1], *k, huber_delta=0.2))
        off = (k[0] + 0.5, k[1] - 0.3, k[2] * 1.1)
        out[f"lm_off{i}"] = np.array(rcf.lm_circle_geometric(xy[:, 0], xy[:, 1], *off, max_iter=25))
    save("kasa_lm", **out)


# 5. RANSAC ------------------------------------------------------------------
def ransac_margin(xy, tol, trials, r_range=None):
    """Replays the reference loop with the reference's own 3-point routine to
    record which trial won and how close any point came to the tolerance."""
    g = np.random.default_rng(42)
    best, best_t, margin = -1, -1, np.inf
    for t in range(trials):
        idx = g.choice(len(xy), size=3, replace=False)
        c = rcf._circle_from_3pts(xy[idx[0]], xy[idx[1]], xy[idx[2]])
        if c is None:
            continue
        if r_range is not None and not (r_range[0] <= c[2] <= r_range[1]):
            continue
        dx = xy[:, 0] - c[0]
        dy = xy[:, 1] - c[1]
        err = np.abs(np.sqrt(dx * dx + dy * dy) - c[2])
        margin = min(margin, float(np.abs(err - tol).min()))
        s = int((err <= tol).sum())
        if s > best:
            best, best_t = s, t
    return best_t, margin


def g_ransac():
    sets = [
        (ring(60, 21), 0.5, 400, None, 0.1), (ring(92, 22, outlier=0.3), 0.5, 4000, None, 0.1),
        (ring(300, 23, outlier=0.3), 0.5, 1024, None, 0.1), (ring(1000, 24, outlier=0.5), 0.3, 512, None, 0.1),
        (ring(8192, 25, outlier=0.3), 0.5, 1024, None, 0.1), (ring(8192, 26), 0.5, 256, None, 0.1),
        (ring(200, 27, outlier=0.3), 0.5, 600, (9.0, 11.0), 0.1), (ring(40, 28, outlier=0.9), 0.05, 300, None, 0.6),
        (ring(5, 29), 0.5, 50, None, 0.1), (ring(3, 30), 0.5, 10, None, 0.1),
        (ring(500, 31, outlier=0.3), 0.5, 200, (20.0, 30.0), 0.1),
    ]
    out = {}
    for i, (xy, tol, trials, rr, frac) in enumerate(sets):
        m = rcf._ransac_circle(xy, inlier_tol=tol, max_trials=trials, min_inliers_frac=frac, r_range=rr)
        out[f"xy{i}"] = xy
        out[f"par{i}"] = np.array([tol, trials, frac, -1 if rr is None else rr[0], -1 if rr is None else rr[1]])
        if m is None:
            out[f"ok{i}"] = np.array(False)
            continue
        bt, margin = ransac_margin(xy, tol, trials, rr)
        out[f"ok{i}"] = np.array(True)
        out[f"model{i}"] = np.array([m["cx"], m["cy"], m["R"]])
        out[f"score{i}"] = np.array(m["score"])
        out[f"mask{i}"] = m["mask"]
        out[f"best_trial{i}"] = np.array(bt)
        out[f"margin{i}"] = np.array(margin)
    save("ransac", **out)


# 6 + 7. unique / hull / hybrid ----------------------------------------------
def slab(n, seed, outlier=0.0, dup=0, sigma=0.05):
    xy = ring(n, seed, outlier=outlier, sigma=sigma)
    g = np.random.default_rng(seed + 1000)
    if dup:
        xy = np.concatenate([xy, xy[g.integers(0, n, dup)]])
    z = g.uniform(-0.9, 0.9, len(xy))
    return np.column_stack([xy, z])


def g_hull_hybrid():
    from scipy.spatial import ConvexHull
    cases = [slab(2000, 41), slab(5000, 42, outlier=0.3), slab(300, 43, dup=60), slab(1200, 44, outlier=0.05),
             slab(64, 45), slab(3000, 46, sigma=0.2), slab(800, 47, outlier=0.6), slab(5, 48), slab(10, 49),
             slab(1500, 50, outlier=0.01)]
    out = {}
    for i, pts in enumerate(cases):
        out[f"pts{i}"] = pts
        res = rcf.fit_circle_hybrid_at_z(pts, 0.0, 0.5, max_trials=800)
        keep = (pts[:, 2] >= -0.5) & (pts[:, 2] <= 0.5)
        uniq = np.unique(pts[keep, :2], axis=0)
        out[f"uniq{i}"] = uniq
        out[f"qhull{i}"] = ConvexHull(uniq).vertices.astype(np.int64) if len(uniq) >= 3 else np.zeros(0, np.int64)
        out[f"ok{i}"] = np.array(res is not None)
        if res is not None:
            out[f"fit{i}"] = np.array([res["center_x"], res["center_y"], res["radius"]])
            out[f"mask{i}"] = res["inlier_mask"]
            out[f"cand{i}"] = res["candidates_xy"]
            assert np.array_equal(res["fit_points_xy"], uniq)
    save("hull_hybrid", **out)


# 8 + 9. boundaries / ovality / residual profile ------------------------------
def g_boundary():
    out = {}
    for i, (xy, bins) in enumerate([(ring(8192, 61), 720), (ring(8192, 62, outlier=0.1), 720), (ring(3000, 63), 90),
                                    (ring(50000, 64), 3600), (ring(400, 65, arc=(1.0, 2.5)), 720)]):
        if len(xy) <= 8192:
            out[f"xy{i}"] = xy
        else:
            out[f"xy{i}_sha"] = np.array(sha(xy))
        b0 = rbd.boundary_by_angle_max(xy, bins=bins)                  # centre from the Kasa fit
        b1 = rbd.boundary_by_angle_max(xy, center=(3.0, -2.0), bins=bins)
        lut = {tuple(p): k for k, p in reversed(list(enumerate(map(tuple, xy))))}
        out[f"bins{i}"] = np.array(bins)
        out[f"idx_auto{i}"] = np.array([lut[tuple(p)] for p in b0], dtype=np.int64)
        out[f"idx_c{i}"] = np.array([lut[tuple(p)] for p in b1], dtype=np.int64)
    for i, xy in enumerate([ring(92, 71), ring(150, 72, sigma=0.2), ring(1500, 73), ring(10, 74)]):
        fit = (3.01, -1.98, 10.02)
        ov = rbd.ovality_and_fourier(xy[:, 0], xy[:, 1], *fit)
        out[f"ov_xy{i}"] = xy
        out[f"ov_fit{i}"] = np.array(fit)
        out[f"ov_scal{i}"] = np.array([ov["Rmax"], ov["Rmin"], ov["ovality_abs"], ov["ovality_pct"], ov["k2_amp"]])
        out[f"ov_dev{i}"] = ov["dev_uniform"]
        # residual profile exactly as gui.py:979-987 writes it
        dx = xy[:, 0] - fit[0]
        dy = xy[:, 1] - fit[1]
        th = (np.arctan2(dy, dx) + 2 * np.pi) % (2 * np.pi)
        dr = np.hypot(dx, dy) - fit[2]
        o = np.argsort(th)
        out[f"rp_theta{i}"] = th[o]
        out[f"rp_dr{i}"] = dr[o]
    save("boundary", **out)


# 10 + 11. analyzer -----------------------------------------------------------
SCALARS = ["z_center", "z_low", "z_high", "n_points_slice", "n_edge", "cx", "cy", "R", "Rmax", "Rmin",
           "x_at_Rmax", "y_at_Rmax", "x_at_Rmin", "y_at_Rmin", "ovality_abs", "ovality_pct", "k2_amp", "cost"]


def g_analyzer():
    out = {}
    pts = make_cylinder(50_000, 20, seed=5, outlier_frac=0.02)
    # points exactly on shared window edges, below the first and above the last window
    pts[10:40, 2] = np.repeat(np.arange(2.0, 32.0, 2.0), 2)
    out["pts_sha"] = np.array(sha(pts))
    out["gen"] = np.array([50_000, 20, 5, 0.02])
    cfg = ran.Config(z_step=2.0, min_points_per_slice=1000)
    az = ran.CylinderAnalyzer(cfg)
    zc = az.calculate_z_centers(pts)
    out["z_centers"] = zc
    counts, first = [], []
    for c in zc:
        m = (pts[:, 2] >= c - 1.0) & (pts[:, 2] <= c + 1.0)
        counts.append(int(m.sum()))
        first.append(np.nonzero(m)[0][:16])
    out["member_counts"] = np.array(counts)
    out["member_first16"] = np.array(first)
    prog = []
    res = az.analyze_with_progress(pts, progress_callback=prog.append)
    out["progress_sorted"] = np.array(sorted(prog))
    out["n_results"] = np.array(len(res))
    out["keys"] = np.array(list(res[0].keys()))
    out["scalars"] = np.array([[r[k] for k in SCALARS] for r in res], dtype=float)
    out["bpts_len"] = np.array([len(r["boundary_points"]) for r in res])
    out["bpts_cat"] = np.concatenate([np.array(r["boundary_points"]) for r in res])
    # one process_slice in full, including the edge frame
    (r0, cand, slc, fit, ov), edge, warn = az.process_slice(pts, zc[3])
    out["ps_scal"] = np.array([r0[k] for k in SCALARS], dtype=float)
    out["ps_edge_cols"] = np.array(list(edge.columns))
    out["ps_edge"] = edge.to_numpy(dtype=float)
    out["ps_slc_len"] = np.array(len(slc))
    sparse = az.process_slice(pts[:500], zc[3])
    out["ps_sparse_msg"] = np.array(sparse[2])
    # configured z range: centres come from the config, not the data
    az2 = ran.CylinderAnalyzer(ran.Config(z_step=4.0, min_points_per_slice=100, z_min=3.0, z_max=29.0))
    res2 = az2.analyze_with_progress(pts)
    out["cfg_z_centers"] = az2.calculate_z_centers(pts)
    out["cfg_scalars"] = np.array([[r[k] for k in SCALARS] for r in res2], dtype=float)
    save("analyzer", **out)


# 12. text loader and CSV writers ------------------------------------------------
TRICKY_LINES = [
    b"# header comment", b"// another", b"; third kind", b"", b"   ", b"1 2 3", b"1.5,2.5,3.5", b" 4.0 , 5.0 ,6.0  ",
    b"7\t8\t9\t10 11 12", b"1e3 -2E-2 +3.25e+1", b".5 5. -.25", b"+1 -0 00012.50", b"1 2", b"1 2 x", b"a b c",
    b"1 2 3 label", b"nan inf -inf", b"NaN Infinity +INF", b"1_000 2 3", b"1__0 2 3", b"0x10 2 3", b"1e999 -1e999 1e-999",
    b"4.9e-324 2.2250738585072014e-308 1.7976931348623157e308", b"1e 2 3", b"1e+ 2 3", b"+-1 2 3", b"--1 2 3",
    b"1..2 2 3", b"1.2.3 2 3", b"- 2 3 4", b". 2 3", b"e5 2 3", b"1,,2,,3", b",1,2,3,", b"1;2;3", b"#1 2 3", b" # 1 2 3",
    b"//1 2 3", b"/ 1 2 3", b"1 2 3 // trailing", b"1 2 3 # trailing", b"\xc2\xa01 2 3\xc2\xa0", b"\xff1 2 3",
    b"\xef\xbb\xbf1 2 3", b"1\x0b2\x0c3", b"\x1c1 2 3", b"\x1c# c", b"1\x002 3 4", b"\xd9\xa1 \xd9\xa2 \xd9\xa3",
    b"0.1 0.2 0.30000000000000004", b"123456789012345678901234567890 1e-5 3", b"9007199254740993 2 3",
    b"1.0000000000000002220446049250313080847263336181640625 2 3", b"3 4 5\r6 7 8\r\n9 10 11", b"12 13 14",
]


def g_data_io():
    from utils import data_io as rio         # (reference)
    import contextlib
    import io
    text = b"\n".join(TRICKY_LINES) + b"\n15 16 17"           # last line without a terminator
    rng = np.random.default_rng(9)
    bulk = rng.normal(0, 100, (2000, 3))
    text += b"\n" + "\n".join(" ".join(repr(float(v)) for v in row) for row in bulk).encode()
    d = tempfile.mkdtemp(prefix="pcfit_io_")
    path = os.path.join(d, "scan.txt")
    with open(path, "wb") as fh:
        fh.write(text)
    pts = rio.load_txt_points(path)
    out = dict(text=np.frombuffer(text, np.uint8), points=pts)
    # writers: the analyzer golden's scan, results and edge frames through save_results
    scan = make_cylinder(50_000, 20, seed=5, outlier_frac=0.02)
    az = ran.CylinderAnalyzer(ran.Config(z_step=2.0, min_points_per_slice=1000))
    results, frames = [], []
    for zc in az.calculate_z_centers(scan)[:4]:
        pack, edge, warn = az.process_slice(scan, zc)
        results.append(pack[0])
        frames.append(edge)
    with contextlib.redirect_stdout(io.StringIO()):
        rio.save_results(results, frames, d)
    out["results_csv"] = np.frombuffer(open(os.path.join(d, "slice_results.csv"), "rb").read(), np.uint8)
    out["points_csv"] = np.frombuffer(open(os.path.join(d, "slice_points.csv"), "rb").read(), np.uint8)
    save("data_io", **out)


# 12. slices beyond one workgroup's LDS: every distinct point, all-point candidates, the profile of 1.py ----
def scratch_script_functions(names):
    """Function objects of the reference's scratch script 1.py.  The script reads an HDF5 file from a fixed
    Windows path while it is imported, so only the wanted function definitions are compiled, as they stand
    in the file, into a namespace that holds what the script's own imports would give them."""
    import ast
    from scipy.spatial import ConvexHull
    from scipy.spatial import QhullError
    path = os.path.join(REF, "1.py")
    tree = ast.parse(open(path, encoding="utf-8").read(), path)
    picked = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    ns = {"np": np, "ConvexHull": ConvexHull, "QhullError": QhullError}
    # (a name defined twice in the script: the later definition wins, as it would on import)
    exec(compile(ast.Module(body=picked, type_ignores=[]), path, "exec"), ns)
    return [ns[n] for n in names]


def g_long_slices():
    from scipy.spatial import ConvexHull
    from oracle.ref_np import make_slab
    _, profile_at_z = scratch_script_functions(["_has_enough_2d_variation", "residual_profile_at_z"])
    out = {}
    every = 97
    cases = [dict(n=5000, seed=71, outlier=0.02, dup=100), dict(n=50000, seed=72, outlier=0.02, dup=500),
             dict(n=20000, seed=73, outlier=0.0, dup=0)]
    out["cases"] = np.array([[c["n"], c["seed"], c["outlier"], c["dup"]] for c in cases])
    out["every"] = np.array(every)
    for i, c in enumerate(cases):
        pts = make_slab(**c)
        out[f"pts_sha{i}"] = np.array(sha(pts))
        keep = (pts[:, 2] >= -0.5) & (pts[:, 2] <= 0.5)
        uniq = np.unique(pts[keep, :2], axis=0)
        out[f"n_unique{i}"] = np.array(len(uniq))
        out[f"uniq_sha{i}"] = np.array(sha(uniq))
        out[f"uniq_sub{i}"] = uniq[::every]
        out[f"qhull{i}"] = ConvexHull(uniq).vertices.astype(np.int64)
        res = rcf.fit_circle_hybrid_at_z(pts, 0.0, 0.5, max_trials=600)
        assert np.array_equal(res["fit_points_xy"], uniq)
        out[f"fit{i}"] = np.array([res["center_x"], res["center_y"], res["radius"]])
        out[f"mask{i}"] = res["inlier_mask"]
        out[f"cand{i}"] = res["candidates_xy"]
        # all distinct points as candidates (circle_fitting.py:196-198 by request)
        res_all = rcf.fit_circle_hybrid_at_z(pts, 0.0, 0.5, max_trials=150, use_hull_vertices_only=False)
        assert np.array_equal(res_all["candidates_xy"], uniq)
        out[f"all_fit{i}"] = np.array([res_all["center_x"], res_all["center_y"], res_all["radius"]])
        out[f"all_mask_sha{i}"] = np.array(sha(res_all["inlier_mask"]))
        out[f"all_mask_sum{i}"] = np.array(int(res_all["inlier_mask"].sum()))
        ov = rbd.ovality_and_fourier(uniq[:, 0], uniq[:, 1], *out[f"all_fit{i}"])
        out[f"all_ov{i}"] = np.array([ov["Rmax"], ov["Rmin"], ov["ovality_abs"], ov["ovality_pct"], ov["k2_amp"]])
        out[f"all_dev_sub{i}"] = ov["dev_uniform"][::every]
        # 1.py:301-345 on the same window
        for tag, flag in (("hull", True), ("all", False)):
            pr = profile_at_z(pts, res, 0.0, 0.5, use_hull_vertices=flag)
            sub = slice(None) if flag else slice(None, None, every)
            out[f"prof_{tag}_n{i}"] = np.array(len(pr["theta"]))
            out[f"prof_{tag}_theta{i}"] = pr["theta"][sub]
            out[f"prof_{tag}_dr{i}"] = pr["delta_r"][sub]
            out[f"prof_{tag}_xy{i}"] = pr["xy_used"][sub]
            out[f"prof_{tag}_xy_sha{i}"] = np.array(sha(pr["xy_used"]))
    save("long_slices", **out)


# 13. the GUI's year export and the comparison's slice matcher (gui.py:1357-1466, 1921-1945) --------------
def gui_year_functions():
    """``export_data_year`` and the file-matching statements of ``compare_years``, compiled out of gui.py
    as they stand.  gui.py imports PyQt6 and VTK at its top, which the container lacks, and both are methods of
    the main window; so the one method definition is compiled as a plain function of ``self`` and the
    statements of the comparison loop up to the sorted table (lines 1921-1945) are compiled as the body of a
    function.  The two dialogs that ``export_data_year`` opens are answered by stand-ins that return what a
    user would enter (year text, directory); everything in between is the reference's own code."""
    import ast
    from datetime import datetime
    import pandas as pd
    path = os.path.join(REF, "gui.py")
    tree = ast.parse(open(path, encoding="utf-8").read(), path)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "CylinderAnalyzerGUI")
    export_def = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "export_data_year")
    compare_def = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "compare_years")
    loop = next(n for n in ast.walk(compare_def) if isinstance(n, ast.For) and isinstance(n.iter, ast.Call)
                and "comparison_files" in ast.unparse(n.iter))
    attempt = loop.body[0]
    assert isinstance(attempt, ast.Try)
    kept = [s for s in attempt.body if not ast.unparse(s).startswith(("ax.", "valid_comparisons"))]
    assert ast.unparse(kept[-1]).startswith("comparison_df_sorted = matching_slices.sort_values")
    collect = ast.parse("found.append((file_path, comparison_year, comparison_df_sorted))").body
    new_loop = ast.For(target=loop.target, iter=loop.iter, body=[ast.Try(body=kept + collect, handlers=attempt.handlers,
                                                                          orelse=[], finalbody=[])], orelse=[])
    matcher = ast.parse("def match_files(self, current_z, current_thickness):\n    found = []\n    return found").body[0]
    matcher.body.insert(1, new_loop)
    module = ast.fix_missing_locations(ast.Module(body=[export_def, matcher], type_ignores=[]))

    class Answers:
        year_text = ""
        directory = ""

    class QInputDialog:
        @staticmethod
        def getText(*a, **k):
            return Answers.year_text, True

    class QFileDialog:
        class Option:
            ShowDirsOnly = 1

        @staticmethod
        def getExistingDirectory(*a, **k):
            return Answers.directory

    ns = {"np": np, "pd": pd, "os": os, "datetime": datetime, "QInputDialog": QInputDialog, "QFileDialog": QFileDialog}
    exec(compile(module, path, "exec"), ns)
    return ns["export_data_year"], ns["match_files"], Answers


def g_year_export():
    export_data_year, match_files, answers = gui_year_functions()

    class Spin:
        def __init__(self, v):
            self.v = v

        def value(self):
            return self.v

    class Bar:
        def __init__(self):
            self.messages = []

        def showMessage(self, m):
            self.messages.append(m)

    class Window:                       # what the two pieces of code read from the main window
        def __init__(self, results, thickness):
            self.current_results = results
            self.slice_thickness = Spin(thickness)
            self.bar = Bar()
            self.comparison_files = []

        def statusBar(self):
            return self.bar

    pts = make_cylinder(30_000, 8, seed=23, outlier_frac=0.02)
    cfg = ran.Config(min_points_per_slice=300, save_plots=False) if "save_plots" in ran.Config.__dataclass_fields__ \
        else ran.Config(min_points_per_slice=300)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        results = ran.CylinderAnalyzer(cfg).analyze_with_progress(pts)
    results = [dict(r) for r in results]
    results.append(dict(results[0], z_center=99.0, boundary_points=[]))     # a slice without boundary points
    d = tempfile.mkdtemp(prefix="pcfit_year_")
    out = {}
    win = Window(results, 2.0)
    answers.year_text, answers.directory = " 2023 ", d
    export_data_year(win)
    assert win.bar.messages and win.bar.messages[-1].startswith("Year 2023 data exported"), win.bar.messages
    bpath = os.path.join(d, "year_2023_boundary_data.csv")
    spath = os.path.join(d, "year_2023_summary.csv")
    out["boundary_csv"] = np.frombuffer(open(bpath, "rb").read(), np.uint8)
    out["summary_csv"] = np.frombuffer(open(spath, "rb").read(), np.uint8)
    # a second year with another thickness, for the matcher's thickness key
    win2 = Window(results[:3], 1.5)
    answers.year_text = "2021"
    export_data_year(win2)
    b2 = os.path.join(d, "year_2021_boundary_data.csv")
    out["boundary_csv_2021"] = np.frombuffer(open(b2, "rb").read(), np.uint8)
    # the inputs: what the window held
    out["n_results"] = np.array(len(results))
    out["scalars"] = np.array([[r["z_center"], r["cx"], r["cy"], r["R"], r.get("ovality_pct", np.nan)] for r in results])
    cnt = np.array([len(r["boundary_points"]) for r in results], np.int64)
    out["boundary_offsets"] = np.concatenate([[0], np.cumsum(cnt)])
    out["boundary_xy"] = np.concatenate([np.asarray(r["boundary_points"], float).reshape(-1, 2) for r in results])
    # the matcher: queries (z, thickness) over the files [2023, summary (invalid format), 2021]
    files = [bpath, spath, b2]
    queries = [(results[3]["z_center"] + 4e-4, 2.0), (results[1]["z_center"], 1.5), (results[3]["z_center"] + 0.5, 2.0),
               (results[2]["z_center"] - 9e-4, 2.0009), (99.0, 2.0)]
    out["queries"] = np.array(queries)
    for q, (z, t) in enumerate(queries):
        win.comparison_files = files
        win.bar.messages.clear()
        found = match_files(win, z, t)
        out[f"match{q}_files"] = np.array([files.index(f) for f, _, _ in found], np.int64)
        for f, year, table in found:
            k = files.index(f)
            assert isinstance(year, str)
            out[f"match{q}_{k}_year"] = np.array(year)
            out[f"match{q}_{k}_index"] = table.index.to_numpy().astype(np.int64)
            out[f"match{q}_{k}_theta"] = table["theta"].to_numpy()
            out[f"match{q}_{k}_delta_r"] = table["delta_r"].to_numpy()
        out[f"match{q}_messages"] = np.array(win.bar.messages)
    save("year_export", **out)


def g_qhull_order():
    """What ``scipy.spatial.ConvexHull(xy).vertices`` (circle_fitting.py:189-190) returns for slices of many
    kinds (oracle.qhull2d.make_case: inputs only): the first vertex, the count and a SHA-256 of the list."""
    from scipy.spatial import ConvexHull
    from oracle.qhull2d import KINDS, make_case
    gen = np.random.default_rng(2024)
    rows, first, count, digest = [], [], [], []
    for r in range(60 * len(KINDS)):
        kind = r % len(KINDS)
        n, seed = int(gen.choice([4, 10, 30, 200, 1500, 8192])), int(gen.integers(1 << 30))
        xy = make_case(KINDS[kind], n, seed)
        if len(xy) < 3:
            continue
        try:
            v = ConvexHull(xy).vertices.astype(np.int64)
        except Exception:           # noqa: BLE001  (Qhull refuses flat input)
            continue
        rows.append([kind, n, seed, len(xy)])
        first.append(int(v[0]))
        count.append(len(v))
        digest.append(sha(v))
    save("qhull_order", kinds=np.array(KINDS), cases=np.array(rows, np.int64), first=np.array(first, np.int64),
         count=np.array(count, np.int64), sha=np.array(digest))


if __name__ == "__main__":
    only = sys.argv[2:]
    if only:
        for name in only:
            globals()["g_" + name]()
        sys.exit(0)
    g_data_io()
    g_choice3()
    g_circle3()
    g_kasa_lm()
    g_ransac()
    g_hull_hybrid()
    g_boundary()
    g_analyzer()
    g_long_slices()
    g_year_export()
    g_qhull_order()
