"""Pins oracle.ref_np (the numpy restatement) to outputs of the real reference
recorded in tests/golden/*.npz.  CPU only."""
import hashlib

import numpy as np
import pytest

from conftest import load_golden, same_cycle, ulp_diff, year_fixture
from oracle import ref_np as R

REL = 1e-12      # same numpy/LAPACK calls as the reference: equal up to BLAS kernel choice


def close(a, b, rel=REL):
    a = np.asarray(a, float)
    b = np.asarray(b, float)
    return np.all(np.abs(a - b) <= rel * np.maximum(1.0, np.abs(b)))


def test_choice3_tables():
    g = load_golden("choice3")
    for key in g.files:
        if key == "meta":
            continue
        if key.startswith("seed"):
            seed = int(key[4:].split("_")[0])
            assert np.array_equal(R.choice3_table(500, 32, seed), g[key]), key
        elif key == "n92_t4000":
            assert np.array_equal(R.choice3_table(92, 4000), g[key])
        else:
            assert np.array_equal(R.choice3_table(int(key[1:]), 64), g[key]), key


def test_circle_3pt():
    g = load_golden("circle3")
    for t, ok, val in zip(g["tri"], g["ok"], g["val"]):
        c = R.circle_3pt(t[0], t[1], t[2])
        assert (c is not None) == bool(ok)
        if ok:
            assert np.all(ulp_diff(np.array(c), val) <= 4)
    assert 0 < g["ok"].sum() < len(g["ok"])


def test_kasa_and_lm():
    g = load_golden("kasa_lm")
    for i in range(12):
        xy = g[f"xy{i}"]
        k = R.kasa_fit(xy[:, 0], xy[:, 1])
        assert close(k, g[f"kasa{i}"]), i
        assert close(R.lm_huber(xy[:, 0], xy[:, 1], *g[f"kasa{i}"]), g[f"lm{i}"], 1e-9), i
        assert close(R.lm_huber(xy[:, 0], xy[:, 1], *g[f"kasa{i}"], huber_delta=0.2), g[f"lm_d02_{i}"], 1e-9), i
        kk = g[f"kasa{i}"]
        off = (kk[0] + 0.5, kk[1] - 0.3, kk[2] * 1.1)
        assert close(R.lm_huber(xy[:, 0], xy[:, 1], *off, max_iter=25), g[f"lm_off{i}"], 1e-9), i
    with pytest.raises(ValueError, match="Need at least 3 points for Pratt fit"):
        R.kasa_fit([0.0, 1.0], [0.0, 1.0])


def test_ransac():
    g = load_golden("ransac")
    seen_fail = False
    for i in range(11):
        xy = g[f"xy{i}"]
        tol, trials, frac, r0, r1 = g[f"par{i}"]
        rr = None if r0 < 0 else (r0, r1)
        m = R.ransac(xy, tol, int(trials), frac, rr, want_margin=True)
        assert (m is not None) == bool(g[f"ok{i}"]), i
        if m is None:
            seen_fail = True
            continue
        assert g[f"margin{i}"] > 1e-9            # fixture is not razor thin
        assert m["score"] == int(g[f"score{i}"])
        assert m["best_trial"] == int(g[f"best_trial{i}"])
        assert np.array_equal(m["mask"], g[f"mask{i}"])
        assert close([m["cx"], m["cy"], m["R"]], g[f"model{i}"])
    assert seen_fail


def test_unique_hull_hybrid():
    g = load_golden("hull_hybrid")
    for i in range(10):
        pts = g[f"pts{i}"]
        keep = (pts[:, 2] >= -0.5) & (pts[:, 2] <= 0.5)
        uniq = R.unique_rows(pts[keep, :2])
        assert np.array_equal(uniq, g[f"uniq{i}"])
        q = g[f"qhull{i}"]
        if len(uniq) >= 3:
            mine = R.hull_ccw(uniq)
            assert mine[0] == 0
            assert same_cycle(mine, q), i           # same vertices, same CCW cyclic order
        start = int(q[0]) if len(q) else None
        res = R.hybrid_fit(pts, 0.0, 0.5, max_trials=800, hull_start=start)
        assert (res is not None) == bool(g[f"ok{i}"]), i
        if res is None:
            continue
        assert np.array_equal(res["candidates_xy"], g[f"cand{i}"])
        assert np.array_equal(res["inlier_mask"], g[f"mask{i}"])
        assert close([res["center_x"], res["center_y"], res["radius"]], g[f"fit{i}"], 1e-9)
        # ... and with Qhull's construction restated instead of its first vertex handed over
        res = R.hybrid_fit(pts, 0.0, 0.5, max_trials=800, hull_start="qhull")
        assert np.array_equal(res["candidates_xy"], g[f"cand{i}"]) and np.array_equal(res["inlier_mask"], g[f"mask{i}"])


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_long_slices_every_distinct_point_and_the_profile_of_the_scratch_script():
    """Windows of 5 000 to 50 000 points: np.unique of the whole window, hull candidates in Qhull's order, all
    distinct points as candidates, ovality over 50 000 candidates, and residual_profile_at_z (1.py:301-345)
    over hull vertices and over all points - against what the reference returned."""
    g = load_golden("long_slices")
    every = int(g["every"])
    for i, (n, seed, outlier, dup) in enumerate(g["cases"]):
        pts = R.make_slab(int(n), int(seed), float(outlier), int(dup))
        assert _sha(pts) == str(g[f"pts_sha{i}"])
        keep = (pts[:, 2] >= -0.5) & (pts[:, 2] <= 0.5)
        uniq = R.unique_rows(pts[keep, :2])
        assert len(uniq) == int(g[f"n_unique{i}"]) and _sha(uniq) == str(g[f"uniq_sha{i}"])
        q = g[f"qhull{i}"]
        res = R.hybrid_fit(pts, 0.0, 0.5, max_trials=600, hull_start=int(q[0]))
        assert np.array_equal(res["candidates_xy"], g[f"cand{i}"])
        assert np.array_equal(res["inlier_mask"], g[f"mask{i}"])
        assert close([res["center_x"], res["center_y"], res["radius"]], g[f"fit{i}"], 1e-9)
        assert np.array_equal(res["fit_points_xy"], uniq)
        allc = R.hybrid_fit(pts, 0.0, 0.5, max_trials=150, use_hull_vertices_only=False)
        assert np.array_equal(allc["candidates_xy"], uniq)
        assert _sha(allc["inlier_mask"]) == str(g[f"all_mask_sha{i}"])
        assert close([allc["center_x"], allc["center_y"], allc["radius"]], g[f"all_fit{i}"], 1e-9)
        ov = R.ovality_fourier(uniq[:, 0], uniq[:, 1], *g[f"all_fit{i}"])
        assert close([ov[k] for k in ("Rmax", "Rmin", "ovality_abs", "ovality_pct", "k2_amp")], g[f"all_ov{i}"], 1e-12)
        assert close(ov["dev_uniform"][::every], g[f"all_dev_sub{i}"], 1e-12)
        fit = dict(center_x=g[f"fit{i}"][0], center_y=g[f"fit{i}"][1], radius=g[f"fit{i}"][2])
        for tag, flag, sub in (("hull", True, slice(None)), ("all", False, slice(None, None, every))):
            pr = R.residual_profile_at_z(pts, fit, 0.0, 0.5, use_hull_vertices=flag, hull_start=int(q[0]))
            assert len(pr["theta"]) == int(g[f"prof_{tag}_n{i}"])
            assert _sha(pr["xy_used"]) == str(g[f"prof_{tag}_xy_sha{i}"])
            assert np.array_equal(pr["xy_used"][sub], g[f"prof_{tag}_xy{i}"])
            assert np.all(ulp_diff(pr["theta"][sub], g[f"prof_{tag}_theta{i}"]) <= 4)
            assert np.all(np.abs(pr["delta_r"][sub] - g[f"prof_{tag}_dr{i}"]) <= 4e-15)


def _ring(n, seed, r=10.0, c=(3.0, -2.0), sigma=0.05):
    gen = np.random.default_rng(seed)
    th = gen.uniform(0, 2 * np.pi, n)
    rr = r + gen.normal(0, sigma, n)
    return np.stack([c[0] + rr * np.cos(th), c[1] + rr * np.sin(th)], axis=1)


def test_boundary_angle_max_and_ovality():
    g = load_golden("boundary")
    for i in range(5):
        if f"xy{i}" in g.files:
            xy = g[f"xy{i}"]
        else:
            xy = _ring(50000, 64)
            assert hashlib.sha256(xy.tobytes()).hexdigest() == str(g[f"xy{i}_sha"])
        bins = int(g[f"bins{i}"])
        assert np.array_equal(R.angle_max_indices(xy, None, bins), g[f"idx_auto{i}"]), i
        assert np.array_equal(R.angle_max_indices(xy, (3.0, -2.0), bins), g[f"idx_c{i}"]), i
    for i in range(4):
        xy = g[f"ov_xy{i}"]
        fit = g[f"ov_fit{i}"]
        ov = R.ovality_fourier(xy[:, 0], xy[:, 1], *fit)
        got = [ov["Rmax"], ov["Rmin"], ov["ovality_abs"], ov["ovality_pct"], ov["k2_amp"]]
        assert close(got, g[f"ov_scal{i}"])
        assert np.allclose(ov["dev_uniform"], g[f"ov_dev{i}"], rtol=0, atol=1e-13)
        th, dr = R.residual_profile(xy, *fit)
        assert np.array_equal(th, g[f"rp_theta{i}"])
        assert np.array_equal(dr, g[f"rp_dr{i}"])


SCALARS = ["z_center", "z_low", "z_high", "n_points_slice", "n_edge", "cx", "cy", "R", "Rmax", "Rmin",
           "x_at_Rmax", "y_at_Rmax", "x_at_Rmin", "y_at_Rmin", "ovality_abs", "ovality_pct", "k2_amp", "cost"]


def analyzer_points():
    pts = R.make_cylinder(50_000, 20, seed=5, outlier_frac=0.02)
    pts[10:40, 2] = np.repeat(np.arange(2.0, 32.0, 2.0), 2)
    return pts


def test_analyzer_slicing_and_results():
    g = load_golden("analyzer")
    pts = analyzer_points()
    assert hashlib.sha256(pts.tobytes()).hexdigest() == str(g["pts_sha"])
    zc, lo, hi = R.slice_windows(pts[:, 2].min(), pts[:, 2].max(), 2.0)
    assert np.array_equal(zc, g["z_centers"])
    offsets, perm = R.bin_slices(pts[:, 2], lo, hi)
    assert np.array_equal(np.diff(offsets), g["member_counts"])
    for s in range(len(zc)):
        assert np.array_equal(perm[offsets[s]:offsets[s] + 16], g["member_first16"][s])
    assert np.diff(offsets).sum() > np.count_nonzero((pts[:, 2] >= lo[0]) & (pts[:, 2] <= hi[-1]))  # shared edges count twice

    res = R.analyze(pts, z_step=2.0, min_points_per_slice=1000)
    ref = g["scalars"]
    assert len(res) == int(g["n_results"])
    assert list(res[0].keys()) == list(g["keys"])
    mine = np.array([[r[k] for k in SCALARS] for r in res], dtype=float)
    exact = [SCALARS.index(k) for k in ("z_center", "z_low", "z_high", "n_points_slice", "n_edge", "cost")]
    assert np.array_equal(mine[:, exact], ref[:, exact])
    # candidate order starts at another hull vertex than Qhull's, so RANSAC samples
    # other triples; where every candidate is an inlier either way the fit agrees tightly
    fit = [SCALARS.index(k) for k in ("cx", "cy", "R")]
    dev = np.abs(mine[:, fit] - ref[:, fit]).max(axis=1) / ref[:, SCALARS.index("R")]
    assert np.median(dev) < 1e-9
    assert np.count_nonzero(dev > 1e-9) <= 4 and dev.max() < 0.05     # slices whose hull holds outliers
    off = 0
    for r, n in zip(res, g["bpts_len"]):
        want = [tuple(p) for p in g["bpts_cat"][off:off + n]]
        assert same_cycle([tuple(p) for p in r["boundary_points"]], want)
        off += n

    zc2, _, _ = R.slice_windows(3.0, 29.0, 4.0)
    assert np.array_equal(zc2, g["cfg_z_centers"])
    res2 = R.analyze(pts, z_step=4.0, z_min=3.0, z_max=29.0, min_points_per_slice=100)
    mine2 = np.array([[r[k] for k in SCALARS] for r in res2], dtype=float)
    assert np.array_equal(mine2[:, exact], g["cfg_scalars"][:, exact])


def test_process_slice_edge_frame_and_skip():
    g = load_golden("analyzer")
    pts = analyzer_points()
    zc = g["z_centers"]
    res, edge, warn = R.process_slice(pts, zc[3])
    assert warn is None
    assert list(g["ps_edge_cols"]) == ["z_center", "x", "y", "radius", "theta", "inlier"]
    assert res["n_points_slice"] == int(g["ps_slc_len"])
    ref_xy = [tuple(p) for p in g["ps_edge"][:, 1:3]]
    assert same_cycle(list(zip(edge["x"], edge["y"])), ref_xy)
    _, _, msg = R.process_slice(pts[:500], zc[3])
    assert msg == str(g["ps_sparse_msg"])


def test_filtered_residual_profile_rules():
    """gui.py:1626-1655: inlier mask when it fits, else the 95 % quantile of |delta_r|; theta ascending, in degrees."""
    g = np.random.default_rng(5)
    th = g.uniform(0, 2 * np.pi, 400)
    r = 10.0 + g.normal(0, 0.05, 400)
    r[::37] += 3.0                                             # gross outliers
    xy = np.stack([3.0 + r * np.cos(th), -2.0 + r * np.sin(th)], 1)
    mask = g.random(400) < 0.7
    td, dr, kept = R.filtered_residual_profile(xy, 3.0, -2.0, 10.0, inlier_mask=mask)
    assert np.array_equal(kept, mask) and len(td) == mask.sum() and np.all(np.diff(td) >= 0) and td.max() < 360.0
    td, dr, kept = R.filtered_residual_profile(xy, 3.0, -2.0, 10.0, inlier_mask=mask[:-1])
    assert kept.sum() == 380 and np.abs(dr).max() < 1.0       # the outliers are gone
    assert np.allclose(np.sort(np.degrees(th[kept])), td)


def test_year_export_files_and_matcher_of_the_gui(tmp_path):
    """gui.py:1357-1466 (export_data_year) and gui.py:1921-1945 (the comparison's slice matcher), run from the
    reference's own source on recorded results: the restatement writes the same bytes and picks the same rows."""
    import io
    import pandas as pd
    g, results = year_fixture()
    for key, res, year, thick in (("boundary_csv", results, "2023", 2.0), ("boundary_csv_2021", results[:3], "2021", 1.5)):
        text = pd.DataFrame(R.year_boundary_rows(res, year, thick)).to_csv(index=False, float_format="%.6f")
        assert text.encode() == g[key].tobytes(), key
    text = pd.DataFrame(R.year_summary_rows(results, "2023", 2.0)).to_csv(index=False, float_format="%.6f")
    assert text.encode() == g["summary_csv"].tobytes()
    files = [g["boundary_csv"], g["summary_csv"], g["boundary_csv_2021"]]
    for q, (z, t) in enumerate(g["queries"]):
        hits = []
        for k, raw in enumerate(files):
            m = R.year_match(pd.read_csv(io.BytesIO(raw.tobytes())), z, t)
            if m is None:
                continue
            hits.append(k)
            assert m[0] == str(g[f"match{q}_{k}_year"])
            assert np.array_equal(m[1].index.to_numpy(), g[f"match{q}_{k}_index"])
            assert np.array_equal(m[1]["theta"].to_numpy(), g[f"match{q}_{k}_theta"], equal_nan=True)
            assert np.array_equal(m[1]["delta_r"].to_numpy(), g[f"match{q}_{k}_delta_r"], equal_nan=True)
        assert hits == g[f"match{q}_files"].tolist(), q
