"""Parity of the HIP path (through the C ABI of libpcfit_hip.so) with the golden
vectors recorded from the reference and with the CPU restatements.  Needs a GPU.

Bars: bit-exact for slice membership and order, sampling tables, RANSAC score /
winning trial / inlier mask / centre, hull vertex sets and order, boundary
indices, histograms, arg-max / arg-min; <= 1e-9 relative for Kasa and LM
(LM_TOL lists the two fixtures whose result the reference's own stopping rule
leaves undetermined beyond ~1e-8); angles within one rounding of 2*pi.
"""
import hashlib
import os

import numpy as np
import pytest

from conftest import load_golden, same_cycle, ulp_diff
from oracle import native as N
from oracle import ref_np as R
from pointcloud_amd import _ffi as F

pytestmark = pytest.mark.gpu

LM_TOL = {1: 1e-7, 10: 1e-4}          # see tests/test_oracle_c.py
KASA_TOL = {9: 1e-9, 10: 1e-8}        # a 22 degree arc and points on a line: ill-conditioned normal equations
THETA_ABS = 2 * np.spacing(2 * np.pi)  # theta = fmod(atan2 + 2pi, 2pi): atan2 differs by an ulp between libms


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


def csr(arrs):
    off = np.zeros(len(arrs) + 1, np.int64)
    np.cumsum([len(a) for a in arrs], out=off[1:])
    cat = np.concatenate(arrs) if arrs else np.zeros((0, 2))
    return np.ascontiguousarray(cat[:, 0]), np.ascontiguousarray(cat[:, 1]), off


# ------------------------------------------------------------------ golden vectors
def test_circle3_golden():
    g = load_golden("circle3")
    out, ok = F.circle3_batch(g["tri"].reshape(-1, 6))
    assert np.array_equal(ok, g["ok"])
    good = g["ok"]
    assert np.array_equal(out[good, :2], g["val"][good, :2])          # LAPACK's 2x2 solve, bit for bit
    assert np.all(ulp_diff(out[good, 2], g["val"][good, 2]) <= 1)     # hypot: libm is not always correctly rounded


def test_kasa_lm_golden_batched():
    g = load_golden("kasa_lm")
    sets = [g[f"xy{i}"] for i in range(12)]
    x, y, off = csr(sets)
    k, st = F.kasa_batch(x, y, off)
    assert np.all(st == 0)
    for i in range(12):
        assert rel(k[i], g[f"kasa{i}"]) < KASA_TOL.get(i, 1e-11), i
    init = np.array([g[f"kasa{i}"] for i in range(12)])
    for key, kw in (("lm", {}), ("lm_d02_", dict(huber_delta=0.2))):
        out, iters = F.lm_batch(x, y, off, init, **kw)
        for i in range(12):
            assert rel(out[i], g[f"{key}{i}"]) < LM_TOL.get(i, 1e-9), (key, i)
    shifted = init + np.array([0.5, -0.3, 0.0])
    shifted[:, 2] = init[:, 2] * 1.1
    out, iters = F.lm_batch(x, y, off, shifted, max_iter=25)
    for i in range(12):
        assert rel(out[i], g[f"lm_off{i}"]) < LM_TOL.get(i, 1e-9), i
        assert iters[i, 0] <= 25


def test_ransac_golden_bit_exact():
    g = load_golden("ransac")
    for i in range(11):
        xy = g[f"xy{i}"]
        tol, trials, frac, r0, r1 = g[f"par{i}"]
        rr = None if r0 < 0 else (r0, r1)
        r = F.ransac_batch(xy[:, 0], xy[:, 1], None, tol, int(trials), frac, rr)
        assert (r["status"][0] == 0) == bool(g[f"ok{i}"]), i
        if r["status"][0]:
            continue
        assert g[f"margin{i}"] > 1e-9
        assert r["score"][0] == int(g[f"score{i}"])
        assert r["best_trial"][0] == int(g[f"best_trial{i}"])
        assert np.array_equal(r["mask"], g[f"mask{i}"])
        assert np.array_equal(r["model"][0, :2], g[f"model{i}"][:2])
        assert ulp_diff(r["model"][0, 2], g[f"model{i}"][2]) <= 1


def test_unique_hull_hybrid_golden():
    g = load_golden("hull_hybrid")
    for i in range(10):
        pts = g[f"pts{i}"]
        off, x, y, perm = F.bin_slices(pts, [-0.5], [0.5])
        keep = (pts[:, 2] >= -0.5) & (pts[:, 2] <= 0.5)
        assert np.array_equal(perm, np.nonzero(keep)[0])
        h = F.unique_hull_batch(x, y, off)
        nu, nh = int(h["n_unique"][0]), int(h["n_hull"][0])
        uniq = g[f"uniq{i}"]
        assert np.array_equal(np.stack([h["ux"][:nu], h["uy"][:nu]], 1), uniq)
        q = g[f"qhull{i}"]
        if len(uniq) >= 3:
            assert h["hull_idx"][0] == 0
            assert same_cycle(h["hull_idx"][:nh], q), i
            assert np.array_equal(h["hull_idx"][:nh], N.hull(uniq))
        # with Qhull's first vertex handed over, everything downstream is the reference's
        start = np.array([int(q[0])]) if len(q) else None
        hy = F.hybrid_batch(x, y, off, max_trials=800, hull_start=start)
        assert (hy["status"][0] == 0) == bool(g[f"ok{i}"]), i
        if hy["status"][0]:
            continue
        assert np.array_equal(np.stack([hy["cand_x"], hy["cand_y"]], 1), g[f"cand{i}"])
        assert np.array_equal(hy["mask"], g[f"mask{i}"])
        assert rel(hy["fit"][0], g[f"fit{i}"]) < 1e-9


def test_hybrid_in_qhulls_order_is_the_reference_without_hints():
    """hull_start="qhull": the library finds the vertex scipy's ConvexHull lists first by itself; candidates,
    inlier mask and fit are then what the reference returned (hull_hybrid.npz, long_slices.npz), and the hull
    walk is ConvexHull(xy).vertices element for element."""
    from pointcloud_amd.utils import circle_fitting as CF
    g = load_golden("hull_hybrid")
    xs, ys, off = [], [], [0]
    for i in range(10):
        pts = g[f"pts{i}"]
        o, x, y, _ = F.bin_slices(pts, [-0.5], [0.5])
        xs.append(x), ys.append(y), off.append(off[-1] + len(x))
        h = F.unique_hull_batch(x, y, o, hull_start="qhull")
        q = g[f"qhull{i}"]
        assert np.array_equal(h["hull_idx"][:int(h["n_hull"][0])], q), i
        res = CF.fit_circle_hybrid_at_z(pts, 0.0, 0.5, max_trials=800, hull_start="qhull")
        assert (res is not None) == bool(g[f"ok{i}"]), i
        if res is not None:
            assert np.array_equal(res["candidates_xy"], g[f"cand{i}"]) and np.array_equal(res["inlier_mask"], g[f"mask{i}"])
            assert rel([res["center_x"], res["center_y"], res["radius"]], g[f"fit{i}"]) < 1e-9
    # the ten slices as one batch, some in Qhull's order, some from their lexicographic minimum, one from a named vertex
    x, y, off = np.concatenate(xs), np.concatenate(ys), np.array(off, np.int64)
    want = np.array([F.HULL_START_QHULL, -1, F.HULL_START_QHULL, int(g["qhull3"][2]), -1, F.HULL_START_QHULL,
                     F.HULL_START_QHULL, -1, F.HULL_START_QHULL, F.HULL_START_QHULL], np.int64)
    h = F.unique_hull_batch(x, y, off, hull_start=want)
    for i in range(10):
        q = g[f"qhull{i}"]
        walk = h["hull_idx"][off[i]:off[i] + int(h["n_hull"][i])]
        if len(q) == 0:
            continue
        first = {F.HULL_START_QHULL: int(q[0]), -1: 0}.get(int(want[i]), int(want[i]))
        assert walk[0] == first and same_cycle(walk, q), i
    # the order as an argument of the call (the default of the Python layer) does the same as naming Qhull's
    # vertex slice by slice, in the batch form and in the scan forms
    hy_named = F.hybrid_batch(x, y, off, max_trials=800, hull_start="qhull", want_profile=False, order="canonical")
    hy_set = F.hybrid_batch(x, y, off, max_trials=800, want_profile=False)
    pts = np.concatenate([np.column_stack([xs[i], ys[i], np.full(len(xs[i]), 2.0 * i)]) for i in range(10)])
    zc = 2.0 * np.arange(10)
    sc = F.scan_hybrid_batch(pts, zc - 0.5, zc + 0.5, max_trials=800, want_profile=False)
    with pytest.raises(ValueError):
        F.hybrid_batch(x, y, off, max_trials=8, order="other")
    assert F.lib().pcfit_hybrid_batch(F._p(x), F._p(y), F._p(off), 10, 0.5, 8, 0.1, None, 1.0, 1, None, 7, 42,
                                      *([None] * 14), 1024) == -1                 # an order that does not exist
    assert b"hull_order" in F.lib().pcfit_last_error()
    for k in ("status", "cand_offsets", "cand_x", "cand_y", "mask", "fit"):
        assert np.array_equal(hy_named[k], hy_set[k], equal_nan=k == "fit"), k
        assert np.array_equal(hy_named[k], sc[k], equal_nan=k == "fit"), k
    for i in range(10):
        c0, c1 = hy_named["cand_offsets"][i], hy_named["cand_offsets"][i + 1]
        if bool(g[f"ok{i}"]):
            assert np.array_equal(np.stack([hy_named["cand_x"][c0:c1], hy_named["cand_y"][c0:c1]], 1), g[f"cand{i}"])
            assert np.array_equal(hy_named["mask"][c0:c1], g[f"mask{i}"])
    # coordinates as a text file with two or three decimals holds them: points exactly in line on hull edges, which
    # Qhull merges away (the restatement's list is pinned to scipy's on such slices in test_qhull_order.py)
    from oracle import qhull2d as Q
    for seed in range(12):
        xy = Q.make_case("rounded", 3000, 900 + seed)
        pts3 = np.column_stack([xy, np.zeros(len(xy))])
        res = CF.fit_circle_hybrid_at_z(pts3, 0.0, 0.5, max_trials=200)
        ref = R.hybrid_fit(pts3, 0.0, 0.5, max_trials=200, hull_start="qhull")
        assert np.array_equal(res["candidates_xy"], ref["candidates_xy"]), seed
        assert np.array_equal(res["inlier_mask"], ref["inlier_mask"]), seed
        assert rel([res["center_x"], res["center_y"], res["radius"]], [ref["center_x"], ref["center_y"], ref["radius"]]) < 1e-9
    g = load_golden("long_slices")
    for i, (n, seed, outlier, dup) in enumerate(g["cases"]):
        pts = R.make_slab(int(n), int(seed), float(outlier), int(dup))
        res = CF.fit_circle_hybrid_at_z(pts, 0.0, 0.5, max_trials=600, hull_start="qhull")
        assert np.array_equal(res["candidates_xy"], g[f"cand{i}"]) and np.array_equal(res["inlier_mask"], g[f"mask{i}"])
        assert rel([res["center_x"], res["center_y"], res["radius"]], g[f"fit{i}"]) < 1e-9


def test_device_replay_of_qhulls_order_against_scipys_lists_and_the_host_replay():
    """pcfit_qhull_start_batch_dev (one workgroup per slice) on the 840 recorded slices: where it answers, the
    vertex is the one scipy listed first (tests/golden/qhull_order.npz); it answers on every slice of the kinds
    in general position, and hands back only slices the host replay is not sure of either, or sure and right."""
    import ctypes as C
    import torch
    from oracle import qhull2d as Q
    g = load_golden("qhull_order")
    kinds = [str(k) for k in g["kinds"]]
    cases = [(kinds[int(k)], Q.make_case(kinds[int(k)], int(n), int(seed))) for k, n, seed, _ in g["cases"]]
    x = np.concatenate([c[1][:, 0] for c in cases])
    y = np.concatenate([c[1][:, 1] for c in cases])
    off = np.concatenate([[0], np.cumsum([len(c[1]) for c in cases])]).astype(np.int64)
    S, total = len(cases), int(off[-1])
    host = F.qhull_order_batch(x, y, off)
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    p = lambda a: C.c_void_p(a.data_ptr())
    dx, dy, doff = t(x), t(y), t(off)
    dnu = t(np.diff(off))
    dnh = t(g["count"].astype(np.int64))
    scratch = torch.empty(total * 48, dtype=torch.uint8, device=dev)
    start = torch.full((S,), -7, dtype=torch.int64, device=dev)
    why = np.zeros(16, np.int64)
    F.check(F.lib().pcfit_debug_qhull_reasons(F._p(why), 1))
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    F.check(F.lib().pcfit_qhull_start_batch_dev(p(dx), p(dy), p(doff), p(dnu), p(dnh), None, S, total, p(scratch),
                                                total * 48, p(start), None, None, None, -1, stream))
    torch.cuda.synchronize()
    got = start.cpu().numpy()
    # the same with the slices' hulls at hand (the kernel then takes the coordinate extremes from among their
    # vertices, and runs in the form that the largest of them asks for), in every form the kernel has
    hh = F.unique_hull_batch(x, y, off)
    hx, hy, hi, hn = t(hh["ux"]), t(hh["uy"]), t(hh["hull_idx"]), t(hh["n_hull"])
    for max_hull in (-1, int(hh["n_hull"].max()), 100, 70):
        again = torch.full((S,), -7, dtype=torch.int64, device=dev)
        F.check(F.lib().pcfit_qhull_start_batch_dev(p(dx), p(dy), p(doff), p(dnu), p(hn), None, S, total, p(scratch),
                                                    total * 48, p(again), p(hx), p(hy), p(hi), max_hull, stream))
        torch.cuda.synchronize()
        again = again.cpu().numpy()
        both = (again != -3) & (got != -3)
        assert both.sum() > 600 and np.array_equal(again[both], got[both]), (max_hull, np.flatnonzero(again != got)[:5])
        if max_hull < 0:                          # the same form: the same slices handed back
            assert np.array_equal(again, got)
    # ... the narrow and the wide ones too, which the library picks by the number of slices to a compute unit
    # (slots + 1 / 3 / 2: workgroups of 64 / 128 / 512 threads); hulls beyond a form's slots are handed back
    try:
        for form in (97, 99, 98, 113, 115, 129, 131, 130, 145, 147, 161, 163, 162, 225, 227, 226, 481, 482):
            os.environ["PCFIT_QHULL_FORM"] = str(form)
            again = torch.full((S,), -7, dtype=torch.int64, device=dev)
            F.check(F.lib().pcfit_qhull_start_batch_dev(p(dx), p(dy), p(doff), p(dnu), p(hn), None, S, total, p(scratch),
                                                        total * 48, p(again), p(hx), p(hy), p(hi), -1, stream))
            torch.cuda.synchronize()
            again = again.cpu().numpy()
            both = (again != -3) & (got != -3)
            assert both.sum() > 500 and np.array_equal(again[both], got[both]), (form, int(both.sum()))
    finally:
        os.environ.pop("PCFIT_QHULL_FORM", None)
    F.check(F.lib().pcfit_debug_qhull_reasons(F._p(why), 1))
    F.check(F.lib().pcfit_qhull_start_batch_dev(p(dx), p(dy), p(doff), p(dnu), p(dnh), None, S, total, p(scratch),
                                                total * 48, p(start), None, None, None, -1, stream))
    torch.cuda.synchronize()
    deferred = got == -3
    assert np.array_equal(got[~deferred], g["first"][~deferred])
    F.check(F.lib().pcfit_debug_qhull_reasons(F._p(why), 1))
    assert int(why[:14].sum()) == int(deferred.sum())          # ([14], [15]: insertions and point visits)
    # handed back: ties, points a few roundoff units above a facet, hulls with points exactly in line on an edge
    # (integer grids, rounded coordinates); on rings, disks, clouds, flat ellipses, arcs and the cylinder's slabs the
    # kernel answers itself
    plain = np.array([c[0] in ("ring", "disk", "gauss", "ringf32", "slab", "ellipse", "arc", "small") for c in cases])
    assert plain.sum() > 400 and deferred[plain].mean() < 0.01, (deferred[plain].mean(), why.tolist())
    # (causes: ties for the farthest point, points a few roundoff units above a new facet, merged hull edges)
    assert why[4] + why[10] + why[12] == deferred.sum(), why.tolist()
    # the library's own route (device first, host for what it hands back) equals the host replay everywhere
    h = F.unique_hull_batch(x, y, off, hull_start="qhull")
    for s_ in range(S):
        nh = int(h["n_hull"][s_])
        if nh >= 3 and host["start"][s_] >= 0:
            assert h["hull_idx"][off[s_]] == host["start"][s_], (s_, cases[s_][0])
        # ... and the whole walk is the list scipy returned, the vertices Qhull merged away left out (rounded
        # coordinates, lattices: the exact predicate of the hull kernel keeps them, the walk in Qhull's order does not)
        assert nh == int(g["count"][s_]) and _sha(h["hull_idx"][off[s_]:off[s_] + nh]) == str(g["sha"][s_]), (s_, cases[s_][0])
    # a hull of more vertices than the kernel has facet slots (every point of a circle): the host replay takes it
    th = np.linspace(0.0, 2 * np.pi, 3000, endpoint=False)
    circ = np.unique(np.stack([5 * np.cos(th) + 0.25, 5 * np.sin(th) - 1.5], 1), axis=0)
    hc = F.unique_hull_batch(circ[:, 0].copy(), circ[:, 1].copy(), None, hull_start="qhull")
    want = F.qhull_order_batch(circ[:, 0].copy(), circ[:, 1].copy(), None)
    assert int(hc["n_hull"][0]) > 2000 and hc["hull_idx"][0] == want["start"][0]
    F.check(F.lib().pcfit_debug_qhull_reasons(F._p(why), 1))
    assert why[8] == 1
    # the slab slices of the hybrid fixtures, long ones included
    for name, count in (("hull_hybrid", 10), ("long_slices", 3)):
        gg = load_golden(name)
        for i in range(count):
            if name == "hull_hybrid":
                uniq = gg[f"uniq{i}"]
            else:
                n, seed, outlier, dup = gg["cases"][i]
                pts = R.make_slab(int(n), int(seed), float(outlier), int(dup))
                uniq = R.unique_rows(pts[(pts[:, 2] >= -0.5) & (pts[:, 2] <= 0.5), :2])
            q = gg[f"qhull{i}"]
            if len(q) < 3:
                continue
            m = len(uniq)
            st1 = torch.full((1,), -7, dtype=torch.int64, device=dev)
            sc = torch.empty(m * 48, dtype=torch.uint8, device=dev)
            keep = [t(uniq[:, 0]), t(uniq[:, 1]), t(np.array([0, m])), t(np.array([m])), t(np.array([len(q)]))]
            F.check(F.lib().pcfit_qhull_start_batch_dev(*[p(a) for a in keep], None, 1, m, p(sc), m * 48, p(st1), None,
                                                        None, None, -1, None))
            torch.cuda.synchronize()
            F.check(F.lib().pcfit_debug_qhull_reasons(F._p(why), 1))
            assert int(st1.item()) == int(q[0]), (name, i, int(st1.item()), why.tolist())


def _ring50k():
    gen = np.random.default_rng(64)
    th = gen.uniform(0, 2 * np.pi, 50000)
    rr = 10.0 + gen.normal(0, 0.05, 50000)
    return np.stack([3.0 + rr * np.cos(th), -2.0 + rr * np.sin(th)], axis=1)


def test_boundary_golden():
    from pointcloud_amd.utils import boundary as B
    from pointcloud_amd.utils import residuals as RS
    g = load_golden("boundary")
    for i in range(5):
        if f"xy{i}" in g.files:
            xy = g[f"xy{i}"]
        else:
            xy = _ring50k()
            assert hashlib.sha256(xy.tobytes()).hexdigest() == str(g[f"xy{i}_sha"])
        bins = int(g[f"bins{i}"])
        assert np.array_equal(B.boundary_by_angle_max(xy, center=(3.0, -2.0), bins=bins), xy[g[f"idx_c{i}"]]), i
        assert np.array_equal(B.boundary_by_angle_max(xy, bins=bins), xy[g[f"idx_auto{i}"]]), i
    for i in range(4):
        xy = g[f"ov_xy{i}"]
        fit = g[f"ov_fit{i}"]
        ov = B.ovality_and_fourier(xy[:, 0], xy[:, 1], *fit)
        want = g[f"ov_scal{i}"]
        got = [ov["Rmax"], ov["Rmin"], ov["ovality_abs"], ov["ovality_pct"]]
        assert rel(got, want[:4]) < 1e-12
        assert abs(ov["k2_amp"] - want[4]) < 1e-12 * np.abs(g[f"ov_dev{i}"]).max() + 1e-9 * want[4]
        assert np.max(np.abs(ov["dev_uniform"] - g[f"ov_dev{i}"])) < 1e-12
        assert np.array_equal(ov["theta"], np.linspace(0, 2 * np.pi, max(1024, len(xy)), endpoint=False))
        th, dr = RS.residual_profile(xy, *fit)
        assert np.max(np.abs(th - g[f"rp_theta{i}"])) <= THETA_ABS
        assert np.max(np.abs(dr - g[f"rp_dr{i}"])) <= 2 * np.spacing(fit[2])
        # the slice view's filtered profile (gui.py:1626-1655): by inlier mask, then by the 95 % quantile of |dr|
        mask = np.random.default_rng(i).random(len(xy)) < 0.8
        for m in (mask, None, mask[:-1]):                            # a mask of the wrong length is ignored
            td, dd, kept = RS.filtered_residual_profile(xy, *fit, inlier_mask=m)
            td_o, dd_o, kept_o = R.filtered_residual_profile(xy, *fit, inlier_mask=m)
            if m is None or len(m) != len(xy):
                # a point whose |dr| sits within rounding of the quantile may fall on either side
                mag = np.abs(R.polar(np.asarray(xy, float), fit[0], fit[1])[1] - fit[2])
                edge = np.abs(mag - np.quantile(mag, 0.95)) <= 4 * np.spacing(fit[2])
                assert np.array_equal(kept[~edge], kept_o[~edge])
                if edge.any():
                    continue
            assert np.array_equal(kept, kept_o)
            # (the conversion to degrees rounds once more on either side)
            assert np.max(np.abs(td - td_o), initial=0.0) <= np.degrees(THETA_ABS) + np.spacing(360.0)
            assert np.max(np.abs(dd - dd_o), initial=0.0) <= 2 * np.spacing(fit[2])
    assert RS.filtered_residual_profile(np.zeros((0, 2)), 0.0, 0.0, 1.0)[0].size == 0


# ------------------------------------------------------------------ against the C restatement
def test_kernels_match_c_oracle_ragged():
    gen = np.random.default_rng(3)
    x, y, _ = R.make_slices(1, 60000, seed=2, outlier_frac=0.3)
    cuts = np.sort(gen.integers(0, 60000, 40))
    off = np.concatenate([[0], cuts, [60000]]).astype(np.int64)
    off[5] = off[4]                     # an empty slice
    off[9] = off[8] + 2                 # a slice of two points
    S = len(off) - 1
    k, st = F.kasa_batch(x, y, off)
    r = F.ransac_batch(x, y, off, max_trials=300)
    fit = np.tile([3.01, -1.98, 10.02], (S, 1))
    res = F.residuals_batch(x, y, off, fit, bins=256, lo=-0.5, hi=0.5, want_radius=True)
    sel = F.angle_max_batch(x, y, off, fit, bins=720)
    for s in range(S):
        sl = slice(off[s], off[s + 1])
        xy = np.stack([x[sl], y[sl]], 1)
        so, ko = N.kasa(x[sl], y[sl])
        assert so == st[s]
        if so == 0:
            assert rel(k[s], ko) < 1e-11
        o = N.ransac(xy, trials=300)
        assert (o["status"], o["score"], o["best_trial"]) == (r["status"][s], r["score"][s], r["best_trial"][s]), s
        if o["status"] == 0:
            assert np.array_equal(o["mask"], r["mask"][sl]) and np.array_equal(o["model"], r["model"][s])
        o = N.residuals(xy, *fit[s], bins=256, lo=-0.5, hi=0.5)
        assert np.array_equal(o["dr"], res["dr"][sl])
        assert np.array_equal(o["hist"], res["hist"][s])
        assert (o["argmax"], o["argmin"]) == (res["argmax"][s], res["argmin"][s])
        if len(xy):
            assert (o["rmax"], o["rmin"]) == (res["rmax"][s], res["rmin"][s])
            assert np.max(np.abs(o["theta"] - res["theta"][sl])) <= THETA_ABS
            assert np.array_equal(res["radius"][sl], N.polar(xy, fit[s][0], fit[s][1])[1])
        assert np.array_equal(N.angle_max(xy, fit[s][0], fit[s][1], 720), sel[s]), s


def test_ransac_interval_test_equals_literal_test_on_awkward_scales():
    # radii and tolerances where (R +- tol)^2 sits near rounding boundaries, tiny and huge scales
    gen = np.random.default_rng(5)
    for scale, tol in ((1.0, 0.5), (1e-3, 2e-4), (1e6, 0.125), (1.0, 0.0), (1.0, 30.0), (7.0, 1e-9)):
        th = gen.uniform(0, 2 * np.pi, 700)
        rr = scale * (1 + gen.normal(0, 0.02, 700))
        xy = np.stack([scale * 3 + rr * np.cos(th), -scale * 2 + rr * np.sin(th)], 1)
        xy[::7] += gen.normal(0, scale, (100, 2))
        r = F.ransac_batch(xy[:, 0], xy[:, 1], None, tol, 200, 0.1)
        o = N.ransac(xy, tol, 200, 0.1)
        assert (o["status"], o["score"], o["best_trial"]) == (r["status"][0], r["score"][0], r["best_trial"][0]), (scale, tol)
        if o["status"] == 0:
            assert np.array_equal(o["mask"], r["mask"])


def test_bin_slices_stable_and_exact():
    pts = R.make_cylinder(300_000, 57, seed=9, outlier_frac=0.05)
    pts[5:119, 2] = np.repeat(np.arange(0.0, 114.0, 2.0), 2)      # points exactly on shared window edges
    pts[200:210, 2] = np.nextafter(np.arange(2.0, 22.0, 2.0), 0)  # one ulp below an edge
    pts[210:220, 2] = np.nextafter(np.arange(2.0, 22.0, 2.0), 1e9)
    pts[300, 2] = np.nan
    zc, lo, hi = R.slice_windows(np.nanmin(pts[:, 2]), np.nanmax(pts[:, 2]), 2.0)
    off, x, y, perm = F.bin_slices(pts, lo, hi)
    o_ref, p_ref = N.bin_slices(pts, lo, hi)
    assert np.array_equal(off, o_ref)
    assert np.array_equal(perm, p_ref)                             # same members, same order
    assert np.array_equal(x, pts[perm, 0]) and np.array_equal(y, pts[perm, 1])
    assert off[-1] > np.count_nonzero((pts[:, 2] >= lo[0]) & (pts[:, 2] <= hi[-1]))
    # many windows, few points each (in one pass, then through groups of 256 windows); and one window only
    zc, lo, hi = R.slice_windows(0.0, 114.0, 0.01)
    o_ref, p_ref = N.bin_slices(pts[:50_000], lo, hi)
    for two_level_from in ("-1", "0"):
        os.environ["PCFIT_BIN_TWO_LEVEL_MIN"] = two_level_from
        try:
            off, x, y, perm = F.bin_slices(pts[:50_000], lo, hi)
            assert np.array_equal(off, o_ref) and np.array_equal(perm, p_ref)
            assert np.array_equal(x, pts[perm, 0]) and np.array_equal(y, pts[perm, 1])
            off, x2, y2, none = F.bin_slices(pts[:50_000], lo, hi, want_perm=False)
            assert none is None and np.array_equal(off, o_ref) and np.array_equal(x2, x) and np.array_equal(y2, y)
            # 700 windows that share edges, z on a grid that hits them: the groups share edges as well
            q = pts[:200_000].copy()
            q[:, 2] = np.round(q[:, 2] * 8.0) / 8.0
            qlo = 0.125 * np.arange(700)
            off, x, y, perm = F.bin_slices(q, qlo, qlo + 0.125)
            oq, pq = N.bin_slices(q, qlo, qlo + 0.125)
            assert np.array_equal(off, oq) and np.array_equal(perm, pq)
            assert np.array_equal(x, q[perm, 0]) and np.array_equal(y, q[perm, 1])
            # outputs too small: the call says how many points there are and writes none
            qhi, o_, x_, y_, t_ = qlo + 0.125, np.zeros(701, np.int64), np.zeros(10), np.zeros(10), np.zeros(1, np.int64)
            with pytest.raises(F.PcfitError, match="outputs hold"):
                F.check(F.lib().pcfit_bin_slices(F._p(q), len(q), F._p(qlo), F._p(qhi), 700, F._p(o_), F._p(x_),
                                                 F._p(y_), None, 10, F._p(t_)))
            assert t_[0] == oq[-1] and not x_.any()
        finally:
            del os.environ["PCFIT_BIN_TWO_LEVEL_MIN"]
    off, x, y, perm = F.bin_slices(pts, [10.0], [10.0])            # zero-width window
    assert np.array_equal(perm, np.nonzero(pts[:, 2] == 10.0)[0])
    # windows that share edges, z on a coarse grid: most points lie on an edge and belong to two windows
    q = pts[:120_000].copy()
    q[:, 2] = np.round(q[:, 2] * 2.0) / 2.0
    zc, lo, hi = R.slice_windows(0.0, 114.0, 0.5)
    off, x, y, perm = F.bin_slices(q, lo, hi)
    o_ref, p_ref = N.bin_slices(q, lo, hi)
    assert np.array_equal(off, o_ref) and np.array_equal(perm, p_ref)
    assert off[-1] > 1.5 * np.count_nonzero((q[:, 2] >= lo[0]) & (q[:, 2] <= hi[-1]))
    assert np.array_equal(x, q[perm, 0]) and np.array_equal(y, q[perm, 1])


def test_resident_scan_with_and_without_its_z_column():
    """A resident scan keeps its z column beside the rows and the count passes of the binning read that; with
    PCFIT_SCAN_Z_COLUMN=0 they read the rows: same offsets, members and order either way, in one level and in two,
    with z on shared window edges and NaN among it (analyzer.py:49-52)."""
    from pointcloud_amd import DeviceScan
    gen = np.random.default_rng(77)
    pts = gen.random((400_000, 3))
    pts[:, 2] = np.round(pts[:, 2] * 700.0 * 4.0) / 4.0 * 0.125
    pts[::997, 2] = np.nan
    lo = 0.125 * np.arange(700)
    hi = lo + 0.125
    got = {}
    for column in ("1", "0"):
        for two_level_from in ("-1", "0"):
            os.environ["PCFIT_SCAN_Z_COLUMN"] = column
            os.environ["PCFIT_BIN_TWO_LEVEL_MIN"] = two_level_from
            try:
                with DeviceScan(pts) as scan:
                    got[column, two_level_from] = F.bin_slices(scan, lo, hi) + (scan.window_counts(lo, hi),)
            finally:
                del os.environ["PCFIT_SCAN_Z_COLUMN"], os.environ["PCFIT_BIN_TWO_LEVEL_MIN"]
    o_ref, p_ref = N.bin_slices(pts, lo, hi)
    for key, (off, x, y, perm, counts) in got.items():
        assert np.array_equal(off, o_ref) and np.array_equal(perm, p_ref), key
        assert np.array_equal(x, pts[perm, 0]) and np.array_equal(y, pts[perm, 1]), key
        assert np.array_equal(counts, np.diff(o_ref)), key


def test_two_level_binning_on_the_device_without_a_read_back():
    """pcfit_bin2_count_dev / pcfit_bin2_scatter_dev: members, order and bytes of the CPU's binning for 2000 windows
    with gaps, 700 windows sharing edges with z on a grid that hits them (also the edges between groups of 64), and
    11 400 windows (groups of 128); with and without perm; a workspace too small for the entries reports it."""
    import ctypes as C
    import torch
    dev = torch.device("cuda", 0)
    L = F.lib()
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    pts = R.make_cylinder(300_000, 100, seed=23, outlier_frac=0.02)           # z in [0, 200]
    pts[7, 2] = np.nan
    grid = pts.copy()
    grid[:, 2] = np.round(grid[:, 2] * 8.0) / 8.0
    cases = ((pts, 0.1 * np.arange(2000), 0.08), (grid, 0.125 * np.arange(700), 0.125),
             (pts[:60_000], 0.01 * np.arange(11_400), 0.01))
    for arr, lo, width in cases:
        hi = lo + width
        o_ref, p_ref = N.bin_slices(arr, lo, hi)
        n, S = len(arr), len(lo)
        geom = np.array([lo[0], hi[0], lo[-1]])
        xyz, dlo, dhi = d(arr), d(lo), d(hi)
        for want_perm in (1, 0):
            wsb = L.pcfit_bin2_workspace_bytes(n, S, want_perm, 0)
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            off = torch.empty(S + 1, dtype=torch.int64, device=dev)
            status = torch.full((1,), -1, dtype=torch.int32, device=dev)
            F.check(L.pcfit_bin2_count_dev(p(xyz), n, p(dlo), p(dhi), S, F._p(geom), want_perm, 0, p(off), p(status),
                                           p(ws), wsb, st))
            assert status.item() == 0 and np.array_equal(off.cpu().numpy(), o_ref)
            total = int(o_ref[-1])
            x = torch.empty(total, dtype=torch.float64, device=dev)
            y = torch.empty_like(x)
            perm = torch.empty(total, dtype=torch.int64, device=dev) if want_perm else None
            F.check(L.pcfit_bin2_scatter_dev(n, S, F._p(geom), want_perm, 0, p(off), p(x), p(y),
                                             p(perm) if want_perm else None, p(ws), wsb, st))
            assert np.array_equal(x.cpu().numpy(), arr[p_ref, 0]) and np.array_equal(y.cpu().numpy(), arr[p_ref, 1])
            if want_perm:
                assert np.array_equal(perm.cpu().numpy(), p_ref)
        # room for 1000 entries only: status 1, every offset zero, the scatter step writes nothing
        wsb = L.pcfit_bin2_workspace_bytes(n, S, 0, 1000)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        F.check(L.pcfit_bin2_count_dev(p(xyz), n, p(dlo), p(dhi), S, F._p(geom), 0, 1000, p(off), p(status), p(ws), wsb,
                                       st))
        assert status.item() == 1 and not off.cpu().numpy().any()
        x = torch.full((16,), 7.0, dtype=torch.float64, device=dev)
        y = x.clone()
        F.check(L.pcfit_bin2_scatter_dev(n, S, F._p(geom), 0, 1000, p(off), p(x), p(y), None, p(ws), wsb, st))
        assert (x == 7.0).all().item() and (y == 7.0).all().item()
    with pytest.raises(F.PcfitError, match="workspace"):
        F.check(L.pcfit_bin2_count_dev(p(xyz), n, p(dlo), p(dhi), S, F._p(geom), 0, 0, p(off), p(status), p(ws), 16, st))


def test_more_windows_than_one_binning_pass_takes():
    """A long scan cut very finely: 40 000 windows (runs of 16 384 inside the library) - members and order as the
    CPU gives them, through bin_slices, the resident scan's window counts and the analyzer."""
    from pointcloud_amd import Config, CylinderAnalyzer
    pts = R.make_cylinder(400_000, 100, seed=19)                  # z in [0, 200]
    lo = 0.005 * np.arange(40_000)
    hi = lo + 0.005                                               # shared edges
    pts[1000:1400, 2] = lo[::100]                                 # points on them, also across the runs' borders
    pts[1400, 2], pts[1401, 2] = lo[16_384], lo[32_768]
    off, x, y, perm = F.bin_slices(pts, lo, hi)
    o_ref, p_ref = N.bin_slices(pts, lo, hi)
    assert np.array_equal(off, o_ref) and np.array_equal(perm, p_ref)
    assert np.array_equal(x, pts[perm, 0]) and np.array_equal(y, pts[perm, 1])
    with F.DeviceScan(pts) as scan:
        assert np.array_equal(scan.window_counts(lo, hi), np.diff(o_ref))
        az = CylinderAnalyzer(Config(z_step=0.01, min_points_per_slice=25))
        res = az.analyze_with_progress(scan)
    zc = az.calculate_z_centers(pts)
    assert len(zc) == 20_000 > 16_384
    counts = np.array([np.count_nonzero(np.abs(pts[:, 2] - c) <= 0.005) for c in zc[::997]])
    got = {r["z_center"]: r for r in res}
    for c, k in zip(zc[::997], counts):
        assert (c in got) == (k >= 25)
        if c in got:
            assert got[c]["n_points_slice"] == k


def test_histograms_and_angular_bins_at_their_upper_bounds():
    """8192 residual bins / 8192 angular bins (more than 64 KB of LDS per workgroup), 4096 bins in the fused
    launch: against the CPU restatement."""
    xy = R.make_slab(30_000, 93, outlier=0.02)[:, :2]
    x, y, off = csr([xy, xy[:5000]])
    fit, _ = F.kasa_batch(x, y, off)
    r = F.residuals_batch(x, y, off, fit, bins=8192, lo=-0.4, hi=0.4)
    sel = F.angle_max_batch(x, y, off, fit, bins=8192)
    for s, pts in enumerate((xy, xy[:5000])):
        o = N.residuals(pts, *fit[s], bins=8192, lo=-0.4, hi=0.4)
        assert np.array_equal(o["hist"], r["hist"][s])
        assert np.array_equal(N.angle_max(pts, fit[s][0], fit[s][1], 8192), sel[s])
    a = _fused_dev(x[:10_000], y[:10_000], np.array([0, 6000, 10_000]), 4096, -0.4, 0.4)
    for s, sl in enumerate((slice(0, 6000), slice(6000, 10_000))):
        o = N.residuals(xy[sl], *a["fit"][s], bins=4096, lo=-0.4, hi=0.4)
        assert np.array_equal(o["hist"], a["hist"][s])


def test_degenerate_slices():
    same = np.tile([[1.5, -2.5]], (10, 1))
    line = np.stack([np.arange(12.0), 2 * np.arange(12.0)], 1)
    axis = np.stack([np.arange(9.0), np.zeros(9)], 1)
    x, y, off = csr([same, line, axis, np.zeros((0, 2)), np.array([[0.0, 0.0], [1.0, 1.0]])])
    k, st = F.kasa_batch(x, y, off)
    assert list(st) == [4, 0, 4, 1, 1] or list(st) == [4, 4, 4, 1, 1]
    assert np.allclose(k[0], [1.5, -2.5, 0.0])                     # minimum-norm solution, as lstsq gives
    assert np.all(np.isnan(k[3])) and np.all(np.isnan(k[4]))
    ok = F.rank_check_batch(x, y, off)
    assert list(ok) == [False, False, False, False, False]
    h = F.hybrid_batch(x, y, off, max_trials=50)
    assert list(h["status"]) == [1, 2, 2, 1, 1]
    r = F.ransac_batch(x, y, off, max_trials=50)
    assert list(r["status"]) == [3, 3, 3, 1, 1] and list(r["score"]) == [-1, -1, -1, -1, -1]


def test_two_runs_are_bitwise_identical():
    x, y, off = R.make_slices(50, 3000, seed=8, outlier_frac=0.2, jitter=0.1)
    a = F.hybrid_batch(x, y, off, max_trials=500)
    b = F.hybrid_batch(x, y, off, max_trials=500)
    for key in ("fit", "cand_x", "cand_y", "mask", "radius", "theta", "ov", "status"):
        assert np.array_equal(a[key], b[key], equal_nan=True), key


@pytest.mark.parametrize("n_slices", [2600, 5000])
def test_device_replay_on_batches_of_many_slices_equals_the_host_replay(n_slices):
    """With more than five slices to a compute unit the replay runs in workgroups of 128 threads, with more than a
    dozen in one wave a slice (the launcher picks by the batch): both against the host replay, slice by slice, on
    the sorted distinct points and on the slices as they come (the route of a resident scan)."""
    from oracle import qhull2d as Q
    gen = np.random.default_rng(n_slices)
    kinds = [k for k in Q.KINDS]
    sets = []
    for s_ in range(n_slices):
        n = int(gen.choice([40, 150, 600, 2500]))
        if s_ % 5 == 0:
            th = gen.uniform(0, 2 * np.pi, n)
            rad = 10 + gen.normal(0, 0.005, n)
            xy = np.unique(np.round(np.stack([3 + rad * np.cos(th), -2 + rad * np.sin(th)], 1), 3), axis=0)
        else:
            xy = Q.make_case(kinds[int(gen.integers(len(kinds)))], n, int(gen.integers(1 << 30)))
        sets.append(xy[gen.permutation(len(xy))])
    def joined(sets):
        off = np.concatenate([[0], np.cumsum([len(v) for v in sets])]).astype(np.int64)
        return (np.ascontiguousarray(np.concatenate([v[:, 0] for v in sets])),
                np.ascontiguousarray(np.concatenate([v[:, 1] for v in sets])), off)
    # (hulls of up to 100 vertices only: the forms of 96 and 128 facet slots are the ones that come in narrow workgroups)
    x, y, off = joined(sets)
    small = F.unique_hull_batch(x, y, off)["n_hull"] <= 100
    sets = [v for v, keep in zip(sets, small) if keep]
    n_slices = len(sets)
    assert n_slices > 2000
    x, y, off = joined(sets)
    why = np.zeros(16, np.int64)
    F.check(F.lib().pcfit_debug_qhull_reasons(F._p(why), 1))
    h = F.unique_hull_batch(x, y, off, hull_start="qhull")
    F.check(F.lib().pcfit_debug_qhull_reasons(F._p(why), 1))
    host = F.qhull_order_batch(h["ux"], h["uy"], off, n_unique=h["n_unique"])
    checked = 0
    for s_ in range(n_slices):
        nh = int(h["n_hull"][s_])
        if nh < 3 or host["start"][s_] < 0:
            continue
        assert np.array_equal(h["hull_idx"][off[s_]:off[s_] + nh], host["walk"][off[s_]:off[s_] + host["n_walk"][s_]]), s_
        checked += 1
    assert checked > 0.8 * n_slices
    assert int(h["n_hull"].max()) + 24 <= 128, int(h["n_hull"].max())      # (or the forms of 256 threads would have run)
    assert int(why[:14].sum()) < 0.25 * n_slices, why.tolist()          # (the kernel answered most of them itself)
    # the slices as they come: hull candidates of the hybrid chain in Qhull's order, with and without the list of
    # distinct points (the second is the replay over unsorted slices with copies)
    a = F.hybrid_batch(x, y, off, max_trials=20, want_profile=False, want_unique=True, order="qhull")
    b = F.hybrid_batch(x, y, off, max_trials=20, want_profile=False, want_unique=False, order="qhull")
    for k in ("status", "cand_offsets", "cand_x", "cand_y", "mask"):
        assert np.array_equal(a[k], b[k]), k
    for s_ in range(0, n_slices, 7):
        nh = int(h["n_hull"][s_])
        c0, c1 = a["cand_offsets"][s_], a["cand_offsets"][s_ + 1]
        if a["status"][s_] == 0 and nh >= 3 and c1 - c0 == nh:
            idx = h["hull_idx"][off[s_]:off[s_] + nh]
            assert np.array_equal(a["cand_x"][c0:c1], h["ux"][off[s_] + idx]), s_


def test_replay_over_slices_with_many_copies_of_every_point():
    """Slices as they come, every point several times over, coordinates rounded so that sets have ties for their
    farthest point: copies of one point then meet in different waves' partial results, and a tie that only one of
    them knows of must survive the join (the slice is handed back, not answered).  Against the route over the sorted
    distinct points, which has no copies."""
    gen = np.random.default_rng(77)
    sets = []
    for s_ in range(int(os.environ.get("COPIES_SLICES", "1200"))):
        n = int(gen.choice([300, 1000, 3000]))
        th = gen.uniform(0, 2 * np.pi, n)
        rad = 10 + gen.normal(0, 0.05, n)
        xy = np.unique(np.round(np.stack([3 + rad * np.cos(th), -2 + rad * np.sin(th)], 1) * 8.0) / 8.0, axis=0)
        xy = np.repeat(xy, int(gen.integers(2, 6)), axis=0)
        sets.append(xy[gen.permutation(len(xy))])
    off = np.concatenate([[0], np.cumsum([len(v) for v in sets])]).astype(np.int64)
    x = np.ascontiguousarray(np.concatenate([v[:, 0] for v in sets]))
    y = np.ascontiguousarray(np.concatenate([v[:, 1] for v in sets]))
    srt = F.hybrid_batch(x, y, off, max_trials=20, want_profile=False, want_unique=True, order="qhull")
    for rep in range(6):                 # (which wave sees which copy changes from run to run)
        raw = F.hybrid_batch(x, y, off, max_trials=20, want_profile=False, want_unique=False, order="qhull")
        for k in ("status", "cand_offsets", "cand_x", "cand_y", "mask"):
            assert np.array_equal(raw[k], srt[k]), (rep, k)


def test_the_walk_keeps_the_vertices_qhull_ends_with_and_no_others():
    """tests/golden/qhull_far_vertex.npz: the exact hull of this slice has 81 vertices, scipy lists 80 (point 875 is
    2.7 roundoff units beyond its neighbours' chord: never outside of a facet by Qhull's threshold).  The library's
    walk in Qhull's order is scipy's list, with the kernel deciding and with the host replay deciding, sorted and as
    the slice comes with copies."""
    g = load_golden("qhull_far_vertex")
    xy, want = g["xy"], g["vertices"]
    x, y = np.ascontiguousarray(xy[:, 0]), np.ascontiguousarray(xy[:, 1])
    plain = F.unique_hull_batch(x, y, None)
    assert int(plain["n_hull"][0]) == len(want) + 1
    try:
        for on_host in ("0", "1"):
            os.environ["PCFIT_QHULL_ON_HOST"] = on_host
            h = F.unique_hull_batch(x, y, None, hull_start="qhull")
            assert np.array_equal(h["hull_idx"][:int(h["n_hull"][0])], want), on_host
            gen = np.random.default_rng(5)
            raw = np.repeat(xy, 2, axis=0)[gen.permutation(2 * len(xy))]
            hy = F.hybrid_batch(np.ascontiguousarray(raw[:, 0]), np.ascontiguousarray(raw[:, 1]), None, max_trials=20,
                                want_profile=False, want_unique=False, order="qhull")
            assert np.array_equal(np.stack([hy["cand_x"], hy["cand_y"]], 1), xy[want]), on_host
    finally:
        os.environ.pop("PCFIT_QHULL_ON_HOST", None)


def test_analyzer_in_qhulls_order_equals_the_reference_run_on_every_slice():
    """The analyzer as it comes (Config.hull_order = "qhull"): its candidates are the reference's
    `xy[hull.vertices]` in the reference's order, so every slice - the three whose hull holds outliers included -
    has the reference's boundary points, fit and metrics (tests/golden/analyzer.npz, recorded from the reference
    itself)."""
    import pointcloud_amd as P
    g = load_golden("analyzer")
    pts = analyzer_points()
    res = P.CylinderAnalyzer(P.Config(z_step=2.0, min_points_per_slice=1000)).analyze_with_progress(pts)
    assert not hasattr(P, "set_hull_order") and not hasattr(F.lib(), "pcfit_set_hull_order")   # no process-wide switch
    assert len(res) == int(g["n_results"])
    mine = np.array([[r[k] for k in SCALARS] for r in res], dtype=float)
    ref = g["scalars"]
    exact = [SCALARS.index(k) for k in ("z_center", "z_low", "z_high", "n_points_slice", "n_edge", "cost",
                                        "x_at_Rmax", "y_at_Rmax", "x_at_Rmin", "y_at_Rmin")]
    assert np.array_equal(mine[:, exact], ref[:, exact])
    off = 0
    for r, n in zip(res, g["bpts_len"]):
        assert np.array_equal(np.array(r["boundary_points"]), g["bpts_cat"][off:off + n])    # same first vertex
        off += n
    fitc = [SCALARS.index(k) for k in ("cx", "cy", "R")]
    dev = np.abs(mine[:, fitc] - ref[:, fitc]).max(axis=1) / ref[:, SCALARS.index("R")]
    assert dev.max() < 1e-9, dev
    for col in ("Rmax", "Rmin", "ovality_abs", "ovality_pct"):
        c = SCALARS.index(col)
        assert rel(mine[:, c], ref[:, c]) < 1e-7, col
    c = SCALARS.index("k2_amp")
    assert np.max(np.abs(mine[:, c] - ref[:, c])) < 1e-9


# ------------------------------------------------------------------ drop-in layer
SCALARS = ["z_center", "z_low", "z_high", "n_points_slice", "n_edge", "cx", "cy", "R", "Rmax", "Rmin",
           "x_at_Rmax", "y_at_Rmax", "x_at_Rmin", "y_at_Rmin", "ovality_abs", "ovality_pct", "k2_amp", "cost"]


def analyzer_points():
    pts = R.make_cylinder(50_000, 20, seed=5, outlier_frac=0.02)
    pts[10:40, 2] = np.repeat(np.arange(2.0, 32.0, 2.0), 2)
    return pts


def test_analyzer_matches_reference_run():
    from pointcloud_amd import Config, CylinderAnalyzer
    g = load_golden("analyzer")
    pts = analyzer_points()
    assert hashlib.sha256(pts.tobytes()).hexdigest() == str(g["pts_sha"])
    assert Config().hull_order == "qhull"
    az = CylinderAnalyzer(Config(z_step=2.0, min_points_per_slice=1000, hull_order="canonical"))
    assert np.array_equal(az.calculate_z_centers(pts), g["z_centers"])
    prog = []
    res = az.analyze_with_progress(pts, progress_callback=prog.append)
    assert sorted(prog) == list(g["progress_sorted"]) and prog == sorted(prog) and prog[-1] == 100
    assert len(res) == int(g["n_results"])
    assert list(res[0].keys()) == list(g["keys"])
    mine = np.array([[r[k] for k in SCALARS] for r in res], dtype=float)
    ref = g["scalars"]
    exact = [SCALARS.index(k) for k in ("z_center", "z_low", "z_high", "n_points_slice", "n_edge", "cost")]
    assert np.array_equal(mine[:, exact], ref[:, exact])
    off = 0
    for r, n in zip(res, g["bpts_len"]):
        want = [tuple(p) for p in g["bpts_cat"][off:off + n]]
        assert same_cycle([tuple(p) for p in r["boundary_points"]], want)   # same hull, other first vertex
        off += n
    # the candidate walk starts at another vertex than Qhull's, so RANSAC draws other triples: the
    # fit agrees to rounding wherever the consensus does not depend on the draw, and differs in the
    # few slices whose hull holds outliers (same slices and sizes as the numpy restatement)
    fitc = [SCALARS.index(k) for k in ("cx", "cy", "R")]
    dev = np.abs(mine[:, fitc] - ref[:, fitc]).max(axis=1) / ref[:, SCALARS.index("R")]
    assert np.median(dev) < 1e-9 and np.count_nonzero(dev > 1e-9) <= 4 and dev.max() < 0.05
    oracle = R.analyze(pts, z_step=2.0, min_points_per_slice=1000)
    theirs = np.array([[r[k] for k in SCALARS] for r in oracle], dtype=float)
    assert rel(mine[:, fitc], theirs[:, fitc]) < 1e-9                      # same start vertex: same fits
    for col in ("Rmax", "Rmin", "ovality_abs", "ovality_pct"):
        c = SCALARS.index(col)
        assert rel(mine[:, c], theirs[:, c]) < 1e-9
    for col in ("x_at_Rmax", "y_at_Rmax", "x_at_Rmin", "y_at_Rmin"):
        c = SCALARS.index(col)
        assert np.array_equal(mine[:, c], theirs[:, c])
    c = SCALARS.index("k2_amp")
    assert np.max(np.abs(mine[:, c] - theirs[:, c])) < 1e-12

    az2 = CylinderAnalyzer(Config(z_step=4.0, min_points_per_slice=100, z_min=3.0, z_max=29.0))
    assert np.array_equal(az2.calculate_z_centers(pts), g["cfg_z_centers"])
    res2 = az2.analyze_with_progress(pts)
    mine2 = np.array([[r[k] for k in SCALARS] for r in res2], dtype=float)
    assert np.array_equal(mine2[:, exact], g["cfg_scalars"][:, exact])


def test_process_slice_contract():
    from pointcloud_amd import Config, CylinderAnalyzer
    g = load_golden("analyzer")
    pts = analyzer_points()
    az = CylinderAnalyzer(Config(z_step=2.0, min_points_per_slice=1000))
    zc = g["z_centers"]
    (res, cand, slc, fit, ov), edge, warn = az.process_slice(pts, zc[3])
    assert warn is None
    assert list(edge.columns) == list(g["ps_edge_cols"])
    assert len(slc) == int(g["ps_slc_len"]) and res["n_points_slice"] == len(slc)
    ref_edge = g["ps_edge"]
    assert same_cycle(list(zip(edge["x"], edge["y"])), [tuple(p) for p in ref_edge[:, 1:3]])
    assert set(ov) == {"Rmax", "Rmin", "ovality_abs", "ovality_pct", "k2_amp", "theta", "dev_uniform"}
    assert len(ov["theta"]) == 1024 and len(ov["dev_uniform"]) == 1024
    # the per-candidate columns are those of the reference for the same candidate, whatever the first vertex
    lut = {tuple(p): k for k, p in enumerate(map(tuple, ref_edge[:, 1:3]))}
    order = np.array([lut[(a, b)] for a, b in zip(edge["x"], edge["y"])])
    if abs(res["cx"] - g["ps_scal"][SCALARS.index("cx")]) < 1e-9:
        assert np.max(np.abs(edge["radius"].to_numpy() - ref_edge[order, 3])) < 1e-8
        assert np.array_equal(edge["inlier"].to_numpy(), ref_edge[order, 5].astype(int))
    out = az.process_slice(pts[:500], zc[3])
    assert out[0] is None and out[1] is None and out[2] == str(g["ps_sparse_msg"])
    with pytest.raises(ValueError, match="Points must be Nx3 array"):
        az.analyze_with_progress(np.zeros((5, 2)))
    with pytest.raises(ValueError, match="No points in specified Z range"):
        CylinderAnalyzer(Config(z_min=1e6, z_max=2e6)).analyze_with_progress(pts)


def test_process_slice_thins_what_it_hands_back_not_what_it_fits():
    """analyzer.py:56-60: a window of more than max_points_for_speed points is thinned (unseeded choice) for the
    caller's plot and for n_points_slice only; the fit sees every point."""
    from pointcloud_amd import Config, CylinderAnalyzer
    pts = R.make_slab(6000, 91, outlier=0.01, half_height=0.99)
    capped = CylinderAnalyzer(Config(min_points_per_slice=100, max_points_for_speed=1500)).process_slice(pts, 0.0)
    whole = CylinderAnalyzer(Config(min_points_per_slice=100)).process_slice(pts, 0.0)
    (res_c, cand_c, slc_c, fit_c, ov_c), edge_c, warn_c = capped
    (res_w, cand_w, slc_w, fit_w, ov_w), edge_w, warn_w = whole
    assert warn_c is None and warn_w is None
    assert res_c["n_points_slice"] == 1500 and res_w["n_points_slice"] == 6000
    assert slc_c.shape == (1500, 2) and slc_w.shape == (6000, 2)
    assert set(map(tuple, slc_c)) <= set(map(tuple, slc_w)) and len(set(map(tuple, slc_c))) == 1500
    assert fit_c == fit_w and np.array_equal(cand_c, cand_w)
    for k in res_w:
        if k != "n_points_slice":
            assert res_c[k] == res_w[k], k
    # too sparse: decided before any fit is run
    none = CylinderAnalyzer(Config(min_points_per_slice=7000)).process_slice(pts, 0.0)
    assert none == (None, None, "SKIP zc=0.000: too sparse (6000 points)")


def test_dropin_function_contracts():
    from pointcloud_amd.utils import circle_fitting as CF
    from pointcloud_amd.utils import boundary as B
    with pytest.raises(ValueError, match="Need at least 3 points for Pratt fit"):
        CF.fit_circle_pratt([0.0, 1.0], [0.0, 1.0])
    with pytest.raises(ValueError, match=r"points_xyz must be \(N,3\)"):
        CF.fit_circle_hybrid_at_z(np.zeros((4, 2)), 0.0, 1.0)
    assert CF.fit_circle_hybrid_at_z(np.zeros((4, 3)) + [0, 0, 5.0], 0.0, 1.0) is None
    assert CF._ransac_circle(np.zeros((2, 2))) is None
    assert CF._circle_from_3pts((0, 0), (1, 1), (2, 2)) is None
    c = CF._circle_from_3pts((1, 0), (0, 1), (-1, 0))
    assert np.allclose(c, (0, 0, 1))
    g = load_golden("hull_hybrid")
    res = CF.fit_circle_hybrid_at_z(g["pts0"], 0.0, 0.5, max_trials=800, hull_start=int(g["qhull0"][0]))
    assert list(res.keys()) == ["center_x", "center_y", "radius", "inlier_mask", "fit_points_xy", "candidates_xy",
                                "z_elevation"]
    assert np.array_equal(res["fit_points_xy"], g["uniq0"]) and np.array_equal(res["candidates_xy"], g["cand0"])
    assert rel([res["center_x"], res["center_y"], res["radius"]], g["fit0"]) < 1e-9
    m = CF._ransac_circle(g["cand0"], max_trials=800)
    assert set(m) == {"score", "cx", "cy", "R", "mask"} and np.array_equal(m["mask"], g["mask0"])
    with pytest.raises(RuntimeError, match="Too few hull vertices"):
        B.boundary_by_convex_hull(np.array([[0, 0], [1, 0], [0, 1], [1, 1.0]]))
    xy = g["uniq1"]
    hull32 = B.boundary_by_convex_hull(xy)
    p32 = R.unique_rows(xy.astype(np.float32).astype(np.float64))
    assert np.array_equal(hull32, p32[R.hull_ccw(p32)])


# ------------------------------------------------------------------ full-size configurations
def test_full_size_fit_residual_pass():
    """10 000 slices x 8192 points: the whole batch on the GPU, 48 slices replayed on the CPU - through the
    three separate calls and through the one launch that bench.py times (pcfit_fit_residuals_batch_dev)."""
    S, n = 10_000, 8192
    x, y, off = R.make_slices(S, n, seed=1, jitter=0.02)
    fit_k, st = F.kasa_batch(x, y, off)
    fit, iters = F.lm_batch(x, y, off, fit_k)
    res = F.residuals_batch(x, y, off, fit, bins=256, lo=-0.5, hi=0.5, want_theta=False, want_dr=False)
    assert np.all(st == 0) and np.all(iters[:, 0] == 100)
    assert np.all(res["hist"].sum(axis=1) == n)                   # 10 sigma range holds every point
    assert abs(np.median(fit[:, 2]) - 10.0) < 0.02
    one = _fused_dev(x, y, off, 256, -0.5, 0.5)                  # the timed launch of bench.py, at its own size
    assert np.all(one["status"] == 0) and np.all(one["iters"][:, 0] == 100)
    assert np.all(one["hist"].sum(axis=1) == n)
    assert rel(one["fit"], fit) < 1e-9 and rel(one["kasa"], fit_k) < 1e-11
    for s in np.random.default_rng(0).choice(S, 48, replace=False):
        sl = slice(off[s], off[s + 1])
        so, ko = N.kasa(x[sl], y[sl])
        assert rel(fit_k[s], ko) < 1e-11
        lo_, it, acc = N.lm(x[sl], y[sl], ko)
        assert rel(fit[s], lo_) < 1e-9
        o = N.residuals(np.stack([x[sl], y[sl]], 1), *fit[s], bins=256, lo=-0.5, hi=0.5, per_point=False)
        assert np.array_equal(o["hist"], res["hist"][s])
        assert (o["argmax"], o["argmin"], o["rmax"], o["rmin"]) == (res["argmax"][s], res["argmin"][s],
                                                                    res["rmax"][s], res["rmin"][s])
        # the one launch: fit against the literal C loop, statistics against the C restatement at its own fit
        assert rel(one["kasa"][s], ko) < 1e-11 and rel(one["fit"][s], lo_) < 1e-9
        o1 = N.residuals(np.stack([x[sl], y[sl]], 1), *one["fit"][s], bins=256, lo=-0.5, hi=0.5, per_point=False)
        assert np.array_equal(o1["hist"], one["hist"][s])
        assert (o1["argmax"], o1["argmin"]) == tuple(one["amm"][s]) and (o1["rmax"], o1["rmin"]) == tuple(one["rmm"][s])


def test_fused_fit_residual_launch_equals_the_separate_kernels():
    import ctypes as C
    import torch
    x, y, off = R.make_slices(300, 4000, seed=4, outlier_frac=0.1, jitter=0.05)
    off = off.copy()
    off[1:] -= np.arange(300) % 7                      # ragged
    off[-1] = off[-2] + 2                             # a two-point slice at the end
    x, y = x[:off[-1]], y[:off[-1]]
    S, H = 300, 128
    k, st = F.kasa_batch(x, y, off)
    fit, it = F.lm_batch(x, y, off, k)
    res = F.residuals_batch(x, y, off, fit, bins=H, lo=-0.4, hi=0.6, want_theta=False, want_dr=False)
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    dx, dy, doff, dedge = t(x), t(y), t(off), t(np.linspace(-0.4, 0.6, H + 1))
    ok_, of_ = torch.empty(S, 3, dtype=torch.float64, device=dev), torch.empty(S, 3, dtype=torch.float64, device=dev)
    oi, os_ = torch.empty(S, 4, dtype=torch.int32, device=dev), torch.empty(S, dtype=torch.int32, device=dev)
    oh = torch.empty(S, H, dtype=torch.int64, device=dev)
    orm, oam = torch.empty(S, 2, dtype=torch.float64, device=dev), torch.empty(S, 2, dtype=torch.int64, device=dev)
    p = lambda a: C.c_void_p(a.data_ptr())
    F.check(F.lib().pcfit_fit_residuals_batch_dev(p(dx), p(dy), p(doff), S, int(np.diff(off).max()), 1.0, 100, 1e-2,
                                                  1e-10, 1e-12, p(ok_), p(of_), p(oi), p(os_), H, p(dedge), p(oh),
                                                  p(orm), p(oam), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    assert np.array_equal(ok_.cpu().numpy()[:-1], k[:-1]) and np.array_equal(os_.cpu().numpy(), st)
    assert np.array_equal(of_.cpu().numpy()[:-1], fit[:-1]) and np.array_equal(oi.cpu().numpy()[:-1], it[:-1])
    assert np.array_equal(oh.cpu().numpy()[:-1], res["hist"][:-1])
    assert np.array_equal(oam.cpu().numpy()[:-1, 0], res["argmax"][:-1])
    assert np.array_equal(orm.cpu().numpy()[:-1, 1], res["rmin"][:-1])
    assert st[-1] == 1 and np.all(np.isnan(of_.cpu().numpy()[-1]))


def _fused_dev(x, y, off, H, lo, hi, want_resid=True, max_iter=100):
    """pcfit_fit_residuals_batch_dev (or pcfit_fit_batch_dev) on device copies of the inputs."""
    import ctypes as C
    import torch
    S = len(off) - 1
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    dx, dy, doff, dedge = t(x), t(y), t(off), t(np.linspace(lo, hi, H + 1))
    ok_, of_ = torch.empty(S, 3, dtype=torch.float64, device=dev), torch.empty(S, 3, dtype=torch.float64, device=dev)
    oi, os_ = torch.empty(S, 4, dtype=torch.int32, device=dev), torch.empty(S, dtype=torch.int32, device=dev)
    oh = torch.zeros(S, H, dtype=torch.int64, device=dev)
    orm, oam = torch.empty(S, 2, dtype=torch.float64, device=dev), torch.empty(S, 2, dtype=torch.int64, device=dev)
    p = lambda a: C.c_void_p(a.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    mx = int(np.diff(off).max())
    if want_resid:
        F.check(F.lib().pcfit_fit_residuals_batch_dev(p(dx), p(dy), p(doff), S, mx, 1.0, max_iter, 1e-2, 1e-10, 1e-12,
                                                      p(ok_), p(of_), p(oi), p(os_), H, p(dedge), p(oh), p(orm),
                                                      p(oam), st))
    else:
        F.check(F.lib().pcfit_fit_batch_dev(p(dx), p(dy), p(doff), S, mx, 1.0, 100, 1e-2, 1e-10, 1e-12, p(ok_),
                                            p(of_), p(oi), p(os_), st))
    torch.cuda.synchronize()
    c = lambda a: a.cpu().numpy()
    return dict(kasa=c(ok_), fit=c(of_), iters=c(oi), status=c(os_), hist=c(oh), rmm=c(orm), amm=c(oam))


def _oracle_resid(pts, fit, edges):
    """The C restatement's residual statistics about `fit` with the given edges (any, not only linspace)."""
    import ctypes as C
    n, H = len(pts), len(edges) - 1
    x, y = np.ascontiguousarray(pts[:, 0]), np.ascontiguousarray(pts[:, 1])
    hist, rmm, amm = np.zeros(H, np.int64), np.zeros(2), np.zeros(2, np.int64)
    N.lib().orc_residuals(x, y, n, fit[0], fit[1], fit[2], None, None, H, np.ascontiguousarray(edges), hist, rmm, amm)
    return hist, rmm, amm


def test_fused_residual_pass_where_the_quick_radius_cannot_decide():
    """The fused launch takes bins and extremes from a quick radius and runs the exact code only where that cannot
    decide.  Cases built to sit on those decisions: histogram ranges that begin / end exactly on a data value (a
    point on the first edge, on the closed last edge, on inner edges), the largest and the smallest radius attained
    twice (first index wins), a point on the centre, edges that are not uniformly spaced, a huge and a tiny bin
    width, slices without any full group of points.  Everything against the C restatement at the kernel's own fit."""
    import ctypes as C
    import torch
    dev = torch.device("cuda", 0)
    gen = np.random.default_rng(2024)

    max_iter = [100]

    def fused(pts, edges):
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        p = lambda a: C.c_void_p(a.data_ptr())
        n, H = len(pts), len(edges) - 1
        dx, dy, doff, dedge = t(pts[:, 0]), t(pts[:, 1]), t(np.array([0, n], np.int64)), t(edges)
        kas, fit = torch.empty(1, 3, dtype=torch.float64, device=dev), torch.empty(1, 3, dtype=torch.float64, device=dev)
        it, st_ = torch.empty(1, 4, dtype=torch.int32, device=dev), torch.empty(1, dtype=torch.int32, device=dev)
        hist = torch.full((1, H), -5, dtype=torch.int64, device=dev)
        rmm, amm = torch.empty(1, 2, dtype=torch.float64, device=dev), torch.empty(1, 2, dtype=torch.int64, device=dev)
        F.check(F.lib().pcfit_fit_residuals_batch_dev(p(dx), p(dy), p(doff), 1, n, 1.0, max_iter[0], 1e-2, 1e-10, 1e-12,
                                                      p(kas), p(fit), p(it), p(st_), H, p(dedge), p(hist), p(rmm), p(amm),
                                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        c = lambda a: a.cpu().numpy()
        if max_iter[0] == 0:                    # the kernel without the LM loop: the Kasa circle is the result
            assert np.array_equal(c(fit), c(kas)) and not c(it).any()
        return c(fit)[0], c(hist)[0], c(rmm)[0], c(amm)[0]

    def check(pts, edges, tag):
        fit, hist, rmm, amm = fused(pts, edges)
        h, r, a = _oracle_resid(pts, fit, edges)
        assert np.array_equal(hist, h), tag
        assert np.array_equal(rmm, r) and np.array_equal(amm, a), tag
        return fit

    # (max_iter = 0 is a kernel of its own - no LM loop, three workgroups to a CU from 4097 points on - with its
    # thresholds for the extremes taken from the first points of the slice)
    for mi, n in [(100, n) for n in (8192, 7000, 4100, 2048, 1024, 1023, 300)] + [(0, n) for n in (8192, 7000, 4100, 1023)]:
        max_iter[0] = mi
        pts = R.make_slab(n, 500 + n, outlier=0.0)[:, :2]
        H = 256
        fit = check(pts, np.linspace(-0.5, 0.5, H + 1), (n, "plain"))
        dr = N.residuals(pts, *fit)["dr"]
        picks = gen.choice(n, 6, replace=False)
        for k in picks:                                            # ranges that start / end on a data value
            check(pts, np.linspace(dr[k], dr[k] + 0.3, H + 1), (n, "first edge on a point"))
            check(pts, np.linspace(dr[k] - 0.3, dr[k], H + 1), (n, "last edge on a point"))
            e = np.linspace(-0.5, 0.5, H + 1)
            b = int(np.clip(np.searchsorted(e, dr[k]), 1, H - 1))
            shift = dr[k] - e[b]
            check(pts, e + shift, (n, "an inner edge next to a point"))
        e = np.linspace(-0.5, 0.5, H + 1)
        e[100] += 3e-4 * (e[1] - e[0])
        check(pts, e, (n, "edges not uniform"))
        check(pts, np.linspace(-1e9, 1e9, 9), (n, "huge bins"))
        check(pts, np.linspace(-1e-9, 1e-9, 513), (n, "tiny bins"))
        # extremes attained twice, and a point (nearly) on the centre
        r = np.hypot(pts[:, 0] - fit[0], pts[:, 1] - fit[1])
        twice = pts.copy()
        i_hi, i_lo = int(np.argmax(r)), int(np.argmin(r))
        j_hi, j_lo = (i_hi + n // 2) % n, (i_lo + n // 3) % n
        if len({i_hi, i_lo, j_hi, j_lo}) == 4:
            twice[j_hi], twice[j_lo] = twice[i_hi], twice[i_lo]
            fit2, hist, rmm, amm = fused(twice, np.linspace(-0.5, 0.5, H + 1))
            h, rr, a = _oracle_resid(twice, fit2, np.linspace(-0.5, 0.5, H + 1))
            assert np.array_equal(hist, h) and np.array_equal(rmm, rr) and np.array_equal(amm, a)
            r2 = np.hypot(twice[:, 0] - fit2[0], twice[:, 1] - fit2[1])
            if np.argmax(r2) in (i_hi, j_hi):
                assert amm[0] == min(i_hi, j_hi)
    # A million units from the origin the last, modelled, LM step is of the size of the centre's own rounding: the
    # centre that is returned may lie twice as far from the last sweep's as the step says, and the bounds on the
    # extreme distances that the residual pass takes over from that sweep have to allow for it (they once did
    # not, and the true smallest radius went unexamined).
    max_iter[0] = 100
    for k in range(60):
        n = int(gen.choice([1420, 2049, 1024, 4096, 8192, 5000]))
        far = R.make_slab(n, 9000 + k, outlier=0.0)[:, :2] + np.array([1.0e6, -2.0e6])
        check(far, np.linspace(-0.5, 0.5, 65), (n, k, "far from the origin"))
    # a symmetric cloud with four points at the origin: the smallest radius is three orders below the rest
    q = R.make_slab(1024, 77)[:, :2] - np.array([3.0, -2.0])
    sym = np.concatenate([q, q * [-1, 1], q * [1, -1], q * [-1, -1]])
    sym[5::1024] = 0.0                                            # a point and its three mirror images
    fit = check(sym, np.linspace(-10.5, 0.5, 257), "points next to the centre")
    assert abs(fit[0]) < 0.01 and abs(fit[1]) < 0.01          # (the Huber-weighted refinement drifts a little)


def test_fit_on_ragged_slices_of_up_to_8192_points():
    """Batches whose longest slice has 4097..8192 points keep half of each slice in LDS: ragged lengths
    (full groups and tails of every size, tiny and empty slices) and outliers beyond the Huber
    threshold (the general form of the sweep)."""
    gen = np.random.default_rng(11)
    lens = np.concatenate([[8192, 4097, 8191, 4096, 5000, 0, 2, 3, 100, 1024, 2048, 6144, 7169],
                           gen.integers(1, 8193, 600)])
    S = len(lens)
    x, y, _ = R.make_slices(1, int(lens.sum()), seed=6, outlier_frac=0.15, jitter=0.05)
    off = np.zeros(S + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    H, lo, hi = 200, -0.4, 0.6
    a = _fused_dev(x, y, off, H, lo, hi)
    b = _fused_dev(x, y, off, H, lo, hi, want_resid=False)
    for key in ("kasa", "fit", "iters", "status"):
        assert np.array_equal(a[key], b[key], equal_nan=True), key             # same passes, with and without residuals
    k, st = F.kasa_batch(x, y, off)                                            # register / streaming kernels
    fit, it = F.lm_batch(x, y, off, k)
    ok = lens >= 3
    assert np.array_equal(a["status"], st) and np.array_equal(a["status"] == 0, ok)
    assert rel(a["kasa"][ok], k[ok]) < 1e-11
    # (with outliers the cost is summed to ~1e-16 relative only, which leaves the point where the
    # reference's own stopping rule triggers undetermined to ~1e-8: the same allowance as LM_TOL)
    assert rel(a["fit"][ok], fit[ok]) < 1e-7
    assert np.all(np.isnan(a["fit"][~ok])) and np.all(np.isnan(a["kasa"][~ok]))
    assert np.mean(a["iters"][ok, 0] == it[ok, 0]) > 0.99                      # literal iteration counts
    # the residual statistics are exact given the fitted circle
    res = F.residuals_batch(x, y, off, a["fit"], bins=H, lo=lo, hi=hi, want_theta=False, want_dr=False)
    assert np.array_equal(a["hist"][ok], res["hist"][ok])
    assert np.array_equal(a["amm"][ok, 0], res["argmax"][ok]) and np.array_equal(a["amm"][ok, 1], res["argmin"][ok])
    assert np.array_equal(a["rmm"][ok, 0], res["rmax"][ok]) and np.array_equal(a["rmm"][ok, 1], res["rmin"][ok])
    for s in list(range(13)) + list(gen.choice(np.arange(13, S), 12, replace=False)):
        if lens[s] < 3:
            continue
        sl = slice(off[s], off[s + 1])
        so, ko = N.kasa(x[sl], y[sl])
        assert rel(a["kasa"][s], ko) < 1e-11
        lo_, _, _ = N.lm(x[sl], y[sl], ko)
        assert rel(a["fit"][s], lo_) < 1e-7, s
    again = _fused_dev(x, y, off, H, lo, hi)
    for key in a:
        assert np.array_equal(a[key], again[key], equal_nan=True), key          # bitwise reproducible
    # max_iter = 0: the kernel without the LM loop (22 points of a thread in registers, 10 in LDS, three workgroups
    # to a CU); the Kasa circle is the fit and the residual statistics are exact about it
    z = _fused_dev(x, y, off, H, lo, hi, max_iter=0)
    assert np.array_equal(z["status"], st) and not z["iters"].any()
    assert np.array_equal(z["fit"], z["kasa"], equal_nan=True) and rel(z["kasa"][ok], k[ok]) < 1e-11
    assert np.all(np.isnan(z["fit"][~ok]))
    res0 = F.residuals_batch(x, y, off, z["fit"], bins=H, lo=lo, hi=hi, want_theta=False, want_dr=False)
    assert np.array_equal(z["hist"][ok], res0["hist"][ok])
    assert np.array_equal(z["amm"][ok, 0], res0["argmax"][ok]) and np.array_equal(z["amm"][ok, 1], res0["argmin"][ok])
    assert np.array_equal(z["rmm"][ok, 0], res0["rmax"][ok]) and np.array_equal(z["rmm"][ok, 1], res0["rmin"][ok])
    for s_ in (0, 1, 2, 4, 12):
        sl = slice(off[s_], off[s_ + 1])
        h, r_, a_ = _oracle_resid(np.stack([x[sl], y[sl]], 1), z["fit"][s_], np.linspace(lo, hi, H + 1))
        assert np.array_equal(z["hist"][s_], h) and np.array_equal(z["rmm"][s_], r_) and np.array_equal(z["amm"][s_], a_)


def _resid_stats_dev(x, y, off, fit, bins, lo, hi):
    """pcfit_residuals_batch_dev without per-point outputs, on device copies (one workgroup per slice whatever its length)."""
    import ctypes as C
    import torch
    S = len(off) - 1
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    p = lambda a: C.c_void_p(a.data_ptr()) if a is not None else None
    dx, dy, doff, dfit = t(x), t(y), t(off), t(fit)
    dedge = t(np.linspace(lo, hi, bins + 1)) if bins else None
    hist = torch.full((S, max(bins, 1)), -7, dtype=torch.int64, device=dev)
    rmm, amm = torch.empty(S, 2, dtype=torch.float64, device=dev), torch.empty(S, 2, dtype=torch.int64, device=dev)
    F.check(F.lib().pcfit_residuals_batch_dev(p(dx), p(dy), p(doff), S, p(dfit), None, None, None, bins, p(dedge),
                                              p(hist) if bins else None, p(rmm), p(amm),
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    c = lambda a: a.cpu().numpy()
    return dict(hist=c(hist)[:, :bins], rmax=c(rmm)[:, 0], rmin=c(rmm)[:, 1], argmax=c(amm)[:, 0], argmin=c(amm)[:, 1])


def test_statistics_only_residual_pass_equals_the_per_point_kernel():
    """Without per-point outputs the residual statistics come from the fused kernel's residual pass about the
    given circles (quick radius, exact code only where it cannot decide, thresholds for the extremes from the first
    points of a slice); with PCFIT_RESIDUALS_EXACT=1 from the kernel that takes the correctly rounded hypot of
    every point.  Same counts, extremes and first indices, on ragged batches of every size class (streamed slices
    beyond 8192 points included), clean and with outliers, for several bin counts and ranges, circles off the data,
    a slice of identical points and points that are not numbers; three slices per batch replayed on the C restatement."""
    gen = np.random.default_rng(5)
    lens = np.array([8192, 0, 1, 2, 3, 255, 256, 257, 1023, 1024, 1025, 4100, 8193, 20001, 7000, 64, 100])
    S = len(lens)
    off = np.zeros(S + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    for outlier in (0.0, 0.2):
        x, y, _ = R.make_slices(1, int(lens.sum()), seed=31, outlier_frac=outlier, jitter=0.05)
        x, y = x.copy(), y.copy()
        x[off[9]:off[10]] = 1.25                                   # a slice of identical points
        y[off[9]:off[10]] = -0.5
        x[off[11] + 17] = np.nan                                   # not a number: counted nowhere, never an extreme
        y[off[14] + 4000] = np.inf
        k, _ = F.kasa_batch(x, y, off)
        fit = np.where(np.isfinite(k), k, 1.0)
        fit[5] = [100.0, 100.0, 5.0]                              # a circle far from its points
        fit[6, :2] = [x[off[6] + 3], y[off[6] + 3]]              # a centre on a point
        for bins, lo, hi in ((256, -0.5, 0.5), (0, 0.0, 1.0), (1, -0.1, 0.1), (64, -20.0, 20.0), (4096, -0.3, 0.3),
                             (200, 0.01, 0.02)):
            os.environ["PCFIT_RESIDUALS_EXACT"] = "1"
            try:
                want = _resid_stats_dev(x, y, off, fit, bins, lo, hi)
            finally:
                del os.environ["PCFIT_RESIDUALS_EXACT"]
            got = _resid_stats_dev(x, y, off, fit, bins, lo, hi)
            host = F.residuals_batch(x, y, off, fit, bins=bins, lo=lo, hi=hi, want_theta=False, want_dr=False)
            tag = (outlier, bins, lo, hi)
            for key in ("rmax", "rmin", "argmax", "argmin") + (("hist",) if bins else ()):
                assert np.array_equal(host[key], want[key], equal_nan=True), (tag, key, "host entry point")
            if bins:
                assert np.array_equal(got["hist"], want["hist"]), tag
            for key in ("rmax", "rmin", "argmax", "argmin"):
                assert np.array_equal(got[key], want[key], equal_nan=True), (tag, key)
            if bins:
                for s_ in (0, 13, int(gen.integers(0, S))):
                    sl = slice(off[s_], off[s_ + 1])
                    h, r_, a_ = _oracle_resid(np.stack([x[sl], y[sl]], 1), fit[s_], np.linspace(lo, hi, bins + 1))
                    assert np.array_equal(got["hist"][s_], h), (tag, s_)
                    if lens[s_]:
                        assert got["argmax"][s_] == a_[0] and got["argmin"][s_] == a_[1] and got["rmax"][s_] == r_[0]


def test_long_slices_on_the_segment_parallel_kernels():
    """Batches with a slice of more than 8192 points are cut into segments, one workgroup each, and combined
    per slice in a fixed order: a mix of lengths around the segment size, empty and tiny slices, outliers."""
    lens = np.array([20000, 0, 8193, 2, 3, 100, 8192, 16384, 100001, 16385, 1, 50000])
    S = len(lens)
    x, y, _ = R.make_slices(1, int(lens.sum()), seed=12, outlier_frac=0.1, jitter=0.05)
    off = np.zeros(S + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    k, st = F.kasa_batch(x, y, off)
    init = np.where(np.isfinite(k), k, 1.0)
    fit, it = F.lm_batch(x, y, off, init)
    fitc = np.tile([3.01, -1.98, 10.02], (S, 1))
    res = F.residuals_batch(x, y, off, fitc, bins=256, lo=-0.5, hi=0.5, want_radius=True)
    sel = F.angle_max_batch(x, y, off, fitc, bins=720)
    for s in range(S):
        sl = slice(off[s], off[s + 1])
        xy = np.stack([x[sl], y[sl]], 1)
        so, ko = N.kasa(x[sl], y[sl])
        assert so == st[s], s
        if so == 0:
            assert rel(k[s], ko) < 1e-11, s
            lo_, _, _ = N.lm(x[sl], y[sl], init[s])
            assert rel(fit[s], lo_) < 1e-7, s                      # (outliers: see LM_TOL)
            assert it[s, 0] == 100 or lens[s] <= 3                # (an exact fit stops on a tiny step)
        o = N.residuals(xy, *fitc[s], bins=256, lo=-0.5, hi=0.5)
        assert np.array_equal(o["dr"], res["dr"][sl])
        assert np.array_equal(o["hist"], res["hist"][s]), s
        assert (o["argmax"], o["argmin"]) == (res["argmax"][s], res["argmin"][s]), s
        if len(xy):
            assert (o["rmax"], o["rmin"]) == (res["rmax"][s], res["rmin"][s])
            assert np.max(np.abs(o["theta"] - res["theta"][sl])) <= THETA_ABS
            assert np.array_equal(res["radius"][sl], N.polar(xy, fitc[s][0], fitc[s][1])[1])
        assert np.array_equal(N.angle_max(xy, fitc[s][0], fitc[s][1], 720), sel[s]), s
    k2, st2 = F.kasa_batch(x, y, off)
    fit2, it2 = F.lm_batch(x, y, off, init)
    assert np.array_equal(k, k2, equal_nan=True) and np.array_equal(fit, fit2, equal_nan=True)   # reproducible
    # the same slices one by one: the short ones then run on the one-workgroup kernels
    for s in (2, 5, 6):
        sl = slice(off[s], off[s + 1])
        k1, _ = F.kasa_batch(x[sl], y[sl], None)
        assert rel(k1[0], k[s]) < 1e-12


def test_full_size_ransac_30pct_outliers():
    """10 000 slices x 8192 points, 30 % outliers, 1024 hypotheses: bit-exact consensus on 24 replayed slices."""
    S, n = 10_000, 8192
    x, y, off = R.make_slices(S, n, seed=2, outlier_frac=0.30)
    r = F.ransac_batch(x, y, off, 0.5, 1024, 0.1)
    assert np.all(r["status"] == 0)
    assert np.all(r["score"] == np.add.reduceat(r["mask"].astype(np.int64), off[:-1]))
    assert r["score"].min() > 0.6 * n
    for s in np.random.default_rng(1).choice(S, 24, replace=False):
        sl = slice(off[s], off[s + 1])
        o = N.ransac(np.stack([x[sl], y[sl]], 1), 0.5, 1024, 0.1)
        assert (o["score"], o["best_trial"]) == (r["score"][s], r["best_trial"][s])
        assert np.array_equal(o["mask"], r["mask"][sl]) and np.array_equal(o["model"], r["model"][s])


# ------------------------------------------------------------------ long slices, scan-sized inputs
def test_unique_hull_on_slices_longer_than_lds():
    """Slices of 20 000 and 70 000 points: every distinct point comes back, sorted (tile sort + rank merges through
    global memory), and the hull - found on what the sector filter leaves - indexes that full set: same vertices in
    the same order as the CPU hull.  The hybrid call without ux / uy reports the filter's survivors in n_unique."""
    gen = np.random.default_rng(11)
    sets = []
    # (points, outlier share, arc): outliers pull the centroid off the centre; a lopsided density does so
    # even more; an open arc leaves sectors empty, and then nothing may be dropped
    for n, outl, arc in ((20_000, 0.0, "full"), (70_000, 0.2, "full"), (3_000, 0.0, "full"), (20_000, 0.02, "full"),
                         (30_000, 0.05, "lopsided"), (12_000, 0.0, "open")):
        th = gen.uniform(0, 2 * np.pi, n)
        if arc == "lopsided":
            th[: n // 2] = gen.uniform(0, 0.5 * np.pi, n // 2)
        if arc == "open":
            th = gen.uniform(0, 1.2 * np.pi, n)
        rr = 10.0 + gen.normal(0, 0.05, n)
        bad = gen.random(n) < outl
        rr[bad] = gen.uniform(0, 14.0, int(bad.sum()))
        xy = np.stack([3 + rr * np.cos(th), -2 + rr * np.sin(th)], 1)
        xy[::50] = xy[1::50][: len(xy[::50])]                 # duplicates
        sets.append(xy)
    x, y, off = csr(sets)
    h = F.unique_hull_batch(x, y, off)
    starts = np.full(len(sets), -1, np.int64)
    for s, xy in enumerate(sets):
        uniq = R.unique_rows(xy)
        o = off[s]
        nu, nh = int(h["n_unique"][s]), int(h["n_hull"][s])
        assert nu == len(uniq) and np.array_equal(np.stack([h["ux"][o:o + nu], h["uy"][o:o + nu]], 1), uniq), s
        assert np.array_equal(h["hull_idx"][o:o + nh], N.hull(uniq)), s
        starts[s] = h["hull_idx"][o + nh // 3]
    # a requested first vertex is honoured on long slices as well
    starts[2] = -1
    h2 = F.unique_hull_batch(x, y, off, hull_start=starts)
    for s in range(len(sets)):
        o, nh = off[s], int(h["n_hull"][s])
        want = h["hull_idx"][o:o + nh]
        k = 0 if starts[s] < 0 else nh // 3
        assert np.array_equal(h2["hull_idx"][o:o + nh], np.concatenate([want[k:], want[:k]])), s
    lean = F.hybrid_batch(x, y, off, max_trials=50, want_unique=False, want_profile=False, order="canonical")
    full = np.array([len(R.unique_rows(xy)) for xy in sets])
    assert np.all(lean["n_unique"] <= full) and lean["n_unique"][1] < 0.3 * full[1] and lean["n_unique"][5] == full[5]
    hy = F.hybrid_batch(x, y, off, max_trials=300, order="canonical")
    assert np.all(hy["status"] == 0)
    for s, xy in enumerate(sets):
        ref = R.hybrid_fit(np.column_stack([xy, np.zeros(len(xy))]), 0.0, 1.0, max_trials=300)
        assert rel(hy["fit"][s], [ref["center_x"], ref["center_y"], ref["radius"]]) < 1e-9
        c0, c1 = hy["cand_offsets"][s], hy["cand_offsets"][s + 1]
        assert np.array_equal(np.stack([hy["cand_x"][c0:c1], hy["cand_y"][c0:c1]], 1), ref["candidates_xy"])
        assert np.array_equal(hy["mask"][c0:c1], ref["inlier_mask"])


def test_many_long_slices_take_the_narrow_hull_launch():
    """From 512 slices on, what the sector filter leaves of a long slice goes to a launch of 256-thread workgroups
    when it is short: the same distinct points, hull vertices and order as the 1024-thread launch that smaller
    batches use, and as the CPU hull on sampled slices. One slice keeps more than the narrow launch holds (an open arc is
    not filtered), two are shorter than the LDS and never filtered."""
    gen = np.random.default_rng(77)
    sets = []
    for s in range(520):
        n = 8300 + int(gen.integers(0, 400))
        th = gen.uniform(0, 2 * np.pi, n)
        rr = 10.0 + gen.normal(0, 0.05, n)
        bad = gen.random(n) < 0.03
        rr[bad] = gen.uniform(0, 14.0, int(bad.sum()))
        sets.append(np.stack([3 + rr * np.cos(th), -2 + rr * np.sin(th)], 1))
    th = gen.uniform(0, 1.2 * np.pi, 9000)
    sets[100] = np.stack([3 + 10 * np.cos(th), -2 + 10.2 * np.sin(th)], 1)      # open arc: nothing dropped
    sets[7] = sets[7][:900]
    sets[300] = sets[300][:5000]
    x, y, off = csr(sets)
    h = F.unique_hull_batch(x, y, off)
    for a, b in ((0, 260), (260, 520)):                         # the same slices in batches below the threshold
        xs, ys, offs = csr(sets[a:b])
        g = F.unique_hull_batch(xs, ys, offs)
        assert np.array_equal(g["n_unique"], h["n_unique"][a:b]) and np.array_equal(g["n_hull"], h["n_hull"][a:b])
        assert np.array_equal(g["status"], h["status"][a:b])
        for s in range(a, b):
            o, og = off[s], offs[s - a]
            nu, nh = int(h["n_unique"][s]), int(h["n_hull"][s])
            assert np.array_equal(h["ux"][o:o + nu], g["ux"][og:og + nu]), s
            assert np.array_equal(h["uy"][o:o + nu], g["uy"][og:og + nu]), s
            assert np.array_equal(h["hull_idx"][o:o + nh], g["hull_idx"][og:og + nh]), s
    assert h["n_unique"][100] == len(R.unique_rows(sets[100])) > 2048
    lean = F.hybrid_batch(x, y, off, max_trials=20, want_unique=False, want_profile=False)
    assert np.median(lean["n_unique"]) < 2048                   # what the filter left is what the narrow launch took
    for s in (0, 7, 100, 300, 519):
        uniq = R.unique_rows(sets[s])
        o = off[s]
        nu, nh = int(h["n_unique"][s]), int(h["n_hull"][s])
        kept = np.stack([h["ux"][o:o + nu], h["uy"][o:o + nu]], 1)
        assert np.array_equal(kept[h["hull_idx"][o:o + nh]], uniq[N.hull(uniq)]), s


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_long_slices_against_the_reference_run():
    """Windows of 5 000, 50 000 and 20 000 points through the drop-in functions, against what the reference
    returned for them (tests/golden/long_slices.npz): fit_points_xy = every distinct point; hull candidates in
    Qhull's order (first vertex handed over), inlier mask and fit; all distinct points as candidates
    (use_hull_vertices_only=False); ovality over that many candidates; residual_profile_at_z (1.py:301-345) over
    hull vertices and over all points."""
    from pointcloud_amd.utils import boundary as B
    from pointcloud_amd.utils import circle_fitting as CF
    from pointcloud_amd.utils import residuals as RS
    g = load_golden("long_slices")
    every = int(g["every"])
    for i, (n, seed, outlier, dup) in enumerate(g["cases"]):
        pts = R.make_slab(int(n), int(seed), float(outlier), int(dup))
        assert _sha(pts) == str(g[f"pts_sha{i}"])
        q = g[f"qhull{i}"]
        res = CF.fit_circle_hybrid_at_z(pts, 0.0, 0.5, max_trials=600, hull_start=int(q[0]))
        assert len(res["fit_points_xy"]) == int(g[f"n_unique{i}"]) and _sha(res["fit_points_xy"]) == str(g[f"uniq_sha{i}"])
        assert np.array_equal(res["candidates_xy"], g[f"cand{i}"])
        assert np.array_equal(res["inlier_mask"], g[f"mask{i}"])
        assert rel([res["center_x"], res["center_y"], res["radius"]], g[f"fit{i}"]) < 1e-9
        allc = CF.fit_circle_hybrid_at_z(pts, 0.0, 0.5, max_trials=150, use_hull_vertices_only=False)
        assert _sha(allc["candidates_xy"]) == str(g[f"uniq_sha{i}"])
        assert int(allc["inlier_mask"].sum()) == int(g[f"all_mask_sum{i}"])
        assert _sha(allc["inlier_mask"]) == str(g[f"all_mask_sha{i}"])
        assert rel([allc["center_x"], allc["center_y"], allc["radius"]], g[f"all_fit{i}"]) < 1e-9
        uniq = res["fit_points_xy"]
        ov = B.ovality_and_fourier(uniq[:, 0], uniq[:, 1], *g[f"all_fit{i}"])
        got = [ov[k] for k in ("Rmax", "Rmin", "ovality_abs", "ovality_pct", "k2_amp")]
        assert rel(got[:2], g[f"all_ov{i}"][:2]) < 4e-16 and rel(got[2:], g[f"all_ov{i}"][2:]) < 1e-9
        assert np.max(np.abs(ov["dev_uniform"][::every] - g[f"all_dev_sub{i}"])) < 1e-12
        fit = dict(center_x=g[f"fit{i}"][0], center_y=g[f"fit{i}"][1], radius=g[f"fit{i}"][2])
        for tag, flag, sub in (("hull", True, slice(None)), ("all", False, slice(None, None, every))):
            pr = RS.residual_profile_at_z(pts, fit, 0.0, 0.5, use_hull_vertices=flag)
            assert len(pr["theta"]) == int(g[f"prof_{tag}_n{i}"])
            assert _sha(pr["xy_used"]) == str(g[f"prof_{tag}_xy_sha{i}"]), (i, tag)
            assert np.max(np.abs(pr["theta"][sub] - g[f"prof_{tag}_theta{i}"])) <= THETA_ABS
            assert np.max(np.abs(pr["delta_r"][sub] - g[f"prof_{tag}_dr{i}"])) <= 4e-15
            # ... and every value, not only the recorded sample, against the CPU restatement
            ref = R.residual_profile_at_z(pts, fit, 0.0, 0.5, use_hull_vertices=flag)
            assert np.array_equal(pr["xy_used"], ref["xy_used"])
            assert np.max(np.abs(pr["theta"] - ref["theta"])) <= THETA_ABS
            assert np.max(np.abs(pr["delta_r"] - ref["delta_r"])) <= 4e-15


def test_distinct_points_of_a_million_point_slice_and_of_many_long_slices():
    """np.unique(axis=0) of one slice of 1.2 M points (8 merge passes) and of 300 slices of 9 000 to 40 000 points
    with many duplicates, bit for bit; hull indices point into the full sorted set."""
    gen = np.random.default_rng(5)
    big = R.make_slab(1_200_000, 81, outlier=0.05, dup=100_000)[:, :2]
    sets = [big]
    for s in range(300):
        n = int(gen.integers(9_000, 40_000))
        xy = R.make_slab(n, 1000 + s, outlier=0.02)[:, :2]
        xy = np.round(xy, 2) if s % 3 == 0 else xy            # a coarse grid: most points repeat
        sets.append(xy)
    sets.append(np.repeat(np.array([[1.0, 2.0]]), 10_000, axis=0))      # one point 10 000 times
    x, y, off = csr(sets)
    h = F.unique_hull_batch(x, y, off)
    for s, xy in enumerate(sets):
        uniq = R.unique_rows(xy)
        o, nu, nh = off[s], int(h["n_unique"][s]), int(h["n_hull"][s])
        assert nu == len(uniq), s
        assert np.array_equal(h["ux"][o:o + nu], uniq[:, 0]) and np.array_equal(h["uy"][o:o + nu], uniq[:, 1]), s
        if s in (0, 1, 2, 3, 150, 300):
            assert np.array_equal(h["hull_idx"][o:o + nh], N.hull(uniq)), s
    assert h["status"][-1] == 1 and h["n_unique"][-1] == 1


def test_one_million_point_scan_through_the_analyzer():
    """The reference's own CPU-runnable case: 1 M points, 200 windows; five windows replayed on the CPU."""
    from pointcloud_amd import Config, CylinderAnalyzer
    pts = R.make_cylinder(1_000_000, 200, seed=0)
    az = CylinderAnalyzer(Config())
    res = az.analyze_with_progress(pts)
    assert len(res) == 200
    zc = az.calculate_z_centers(pts)
    assert np.array_equal([r["z_center"] for r in res], zc)
    for s in (0, 57, 123, 198, 199):
        want, _, warn = R.process_slice(pts, zc[s], hull_start="qhull")        # hulls walked as scipy's ConvexHull does
        assert warn is None
        got = res[s]
        assert got["n_points_slice"] == want["n_points_slice"] and got["n_edge"] == want["n_edge"]
        assert got["boundary_points"] == want["boundary_points"]
        for key in ("cx", "cy", "R", "Rmax", "Rmin", "ovality_abs", "ovality_pct"):
            assert abs(got[key] - want[key]) <= 1e-9 * max(1.0, abs(want[key])), (s, key)
        for key in ("x_at_Rmax", "y_at_Rmax", "x_at_Rmin", "y_at_Rmin"):
            assert got[key] == want[key]
        assert abs(got["k2_amp"] - want["k2_amp"]) < 1e-12


def test_boundary_and_residual_statistics_in_one_sweep():
    """pcfit_boundary_residuals_batch: the angular-bin boundary and the residual statistics of every slice from one
    read of it.  Indices, counts and extremes equal those of the two separate calls (their exact kernels) and of the
    C restatement - slice lengths around the kernel's shape boundaries, rings, outliers, open arcs (empty bins),
    duplicates and a lattice (equal radii: the first index wins), a slice a million units from the origin, points in
    ascending order of radius (every point a record breaker: the candidate list overflows and the exact passes
    take over), a point on the centre, NaN coordinates."""
    gen = np.random.default_rng(515)
    sets = []
    for n in (0, 1, 2, 3, 255, 256, 257, 1023, 1024, 1025, 4095, 4096, 4097, 8193, 20001, 50000):
        kind = ["ring", "outliers", "arc", "dups", "far"][len(sets) % 5]
        th = gen.uniform(0, 1.1 * np.pi if kind == "arc" else 2 * np.pi, n)
        rr = 10.0 + gen.normal(0, 0.05, n)
        if kind == "outliers":
            bad = gen.random(n) < 0.2
            rr[bad] = gen.uniform(0, 14.0, int(bad.sum()))
        c = np.array([1e6, -2e6]) if kind == "far" else np.array([3.0, -2.0])
        xy = c + np.stack([rr * np.cos(th), rr * np.sin(th)], 1)
        if kind == "dups" and n > 4:
            xy[n // 2:] = xy[: n - n // 2]
        sets.append((xy, c))
    gx, gy = np.meshgrid(np.arange(-20.0, 21.0), np.arange(-20.0, 21.0))
    sets.append((np.stack([gx.ravel(), gy.ravel()], 1), np.array([0.0, 0.0])))          # lattice: ties, a point on the centre
    th = np.linspace(0, 40 * np.pi, 30000)
    sets.append((np.stack([(5 + th / 30) * np.cos(th), (5 + th / 30) * np.sin(th)], 1), np.array([0.0, 0.0])))   # spiral outwards
    bad = sets[3][0].copy()
    sets.append((np.concatenate([bad, [[np.nan, 1.0], [2.0, np.nan]]]), np.array([3.0, -2.0])))
    x, y, off = csr([a for a, _ in sets])
    fit = np.array([[c[0] + 0.01, c[1] - 0.02, 10.0] for _, c in sets])
    for bins, H, lo, hi in ((720, 256, -0.5, 0.5), (90, 64, -12.0, 25.0), (4096, 0, 0.0, 1.0), (1, 7, -0.2, 0.2)):
        both = F.boundary_residuals_batch(x, y, off, fit, bins=bins, hist_bins=H, lo=lo, hi=hi)
        os.environ["PCFIT_ANGLE_MAX_EXACT"] = "1"
        os.environ["PCFIT_RESIDUALS_EXACT"] = "1"
        try:
            sel = F.angle_max_batch(x, y, off, fit, bins=bins)
            res = F.residuals_batch(x, y, off, fit, bins=H, lo=lo, hi=hi, want_theta=False, want_dr=False)
        finally:
            del os.environ["PCFIT_ANGLE_MAX_EXACT"], os.environ["PCFIT_RESIDUALS_EXACT"]
        assert np.array_equal(both["sel"], sel), bins
        if H:
            assert np.array_equal(both["hist"], res["hist"]), H
        for k in ("argmax", "argmin", "rmax", "rmin"):
            assert np.array_equal(both[k], res[k]), k
        for s in (3, 9, 13, len(sets) - 3, len(sets) - 2):
            xy = sets[s][0]
            assert np.array_equal(N.angle_max(xy, fit[s][0], fit[s][1], bins), both["sel"][s]), (bins, s)
            if H:
                o = N.residuals(xy, *fit[s], bins=H, lo=lo, hi=hi, per_point=False)
                assert np.array_equal(o["hist"], both["hist"][s]), (H, s)
                assert (o["argmax"], o["argmin"]) == (both["argmax"][s], both["argmin"][s])
    # the boundary alone takes the same single sweep
    assert np.array_equal(F.angle_max_batch(x, y, off, fit, bins=720),
                          F.boundary_residuals_batch(x, y, off, fit, bins=720)["sel"])


def test_scan_sized_boundary_and_histogram_pass():
    """20 M points in 400 windows (50 000 per window): binning, Kasa + LM from global memory, angular-bin
    boundary and residual histograms; eight windows replayed on the CPU."""
    n, S = 20_000_000, 400
    pts = R.make_cylinder(n, S, seed=3, outlier_frac=0.1)
    zc, lo, hi = R.slice_windows(0.0, S * 2.0, 2.0)
    off, x, y, _ = F.bin_slices(pts, lo, hi, want_perm=False)
    z = pts[:, 2]
    kasa, st = F.kasa_batch(x, y, off)
    fit, iters = F.lm_batch(x, y, off, kasa)
    sel = F.angle_max_batch(x, y, off, fit, bins=720)
    res = F.residuals_batch(x, y, off, fit, bins=256, lo=-0.5, hi=0.5, want_theta=False, want_dr=False)
    both = F.boundary_residuals_batch(x, y, off, fit, bins=720, hist_bins=256, lo=-0.5, hi=0.5)    # one read of every window
    assert np.array_equal(both["sel"], sel) and np.array_equal(both["hist"], res["hist"])
    assert np.array_equal(both["argmax"], res["argmax"]) and np.array_equal(both["rmin"], res["rmin"])
    assert off[-1] == np.count_nonzero((z >= lo[0]) & (z <= hi[-1])) + np.count_nonzero(np.isin(z, lo[1:]))
    for s in np.random.default_rng(2).choice(S, 8, replace=False):
        m = R.slice_members(z, lo[s], hi[s])
        assert off[s + 1] - off[s] == m.size
        assert np.array_equal(x[off[s]:off[s + 1]], pts[m, 0]) and np.array_equal(y[off[s]:off[s + 1]], pts[m, 1])
        xy = pts[m, :2]
        so, ko = N.kasa(xy[:, 0], xy[:, 1])
        assert rel(kasa[s], ko) < 1e-11
        lo_, _, _ = N.lm(xy[:, 0], xy[:, 1], ko)
        assert rel(fit[s], lo_) < LM_TOL[1]                     # 10 % outliers: noise-limited stopping rule
        assert np.array_equal(N.angle_max(xy, fit[s][0], fit[s][1], 720), sel[s])
        o = N.residuals(xy, *fit[s], bins=256, lo=-0.5, hi=0.5, per_point=False)
        assert np.array_equal(o["hist"], res["hist"][s])
        assert (o["argmax"], o["argmin"]) == (res["argmax"][s], res["argmin"][s])
    # the 400 windows again through two groups of windows: the same bytes
    os.environ["PCFIT_BIN_TWO_LEVEL_MIN"] = "0"
    try:
        off_b, x_b, y_b, _ = F.bin_slices(pts, lo, hi, want_perm=False)
    finally:
        del os.environ["PCFIT_BIN_TWO_LEVEL_MIN"]
    assert np.array_equal(off_b, off) and np.array_equal(x_b, x) and np.array_equal(y_b, y)
    # the same scan in 200 windows of twice the width: up to 256 windows the scatter orders its tiles in LDS
    lo2, hi2 = lo[0::2], hi[1::2]
    off2, x2, y2, perm2 = F.bin_slices(pts, lo2, hi2)
    assert off2[-1] == np.count_nonzero((z >= lo2[0]) & (z <= hi2[-1])) + np.count_nonzero(np.isin(z, lo2[1:]))
    assert np.array_equal(x2, pts[perm2, 0]) and np.array_equal(y2, pts[perm2, 1])
    for s in (0, 77, 199):
        assert np.array_equal(perm2[off2[s]:off2[s + 1]], R.slice_members(z, lo2[s], hi2[s]))


def test_scan_level_call_matches_bin_then_hybrid():
    """pcfit_scan_hybrid_batch (scan uploaded once, binned and fitted on the device) returns what
    pcfit_bin_slices followed by pcfit_hybrid_batch returns, bit for bit."""
    from pointcloud_amd.synth import make_cylinder
    pts = make_cylinder(300_000, 40, seed=11, outlier_frac=0.05)
    zc = np.arange(pts[:, 2].min() + 1.0, pts[:, 2].max() - 1.0 + 1e-9, 2.0)
    off, x, y, _ = F.bin_slices(pts, zc - 1.0, zc + 1.0, want_perm=False)
    a = F.hybrid_batch(x, y, off, want_profile=True)
    b = F.scan_hybrid_batch(pts, zc - 1.0, zc + 1.0, want_profile=True)
    assert np.array_equal(off, b["offsets"])
    for k in ("status", "cand_offsets", "cand_x", "cand_y", "mask", "radius", "theta", "argmax", "argmin"):
        assert np.array_equal(a[k], b[k]), k
    # (without the list of distinct points the hull kernel drops interior points first: fewer "unique" points)
    assert np.all(b["n_unique"] <= a["n_unique"]) and np.all(b["n_unique"] >= np.diff(b["cand_offsets"]))
    assert np.array_equal(a["fit"], b["fit"], equal_nan=True)
    assert np.array_equal(a["ov"], b["ov"], equal_nan=True)
    K = min(a["grid_len"], b["grid_len"])
    assert np.array_equal(a["dev_uniform"][:, :K], b["dev_uniform"][:, :K], equal_nan=True)
    # candidate arrays that are too small are grown and the call repeated
    c = F.scan_hybrid_batch(pts, zc - 1.0, zc + 1.0, cand_capacity=7)
    assert np.array_equal(c["cand_x"], a["cand_x"]) and np.array_equal(c["fit"], a["fit"], equal_nan=True)


def test_year_export_equals_the_files_the_gui_wrote(tmp_path):
    """export_year_data / match_year_slices on the recorded window state of tests/golden/year_export.npz:
    the bytes the reference's own export_data_year (gui.py:1357-1466) wrote and the rows its comparison loop
    (gui.py:1921-1945) picked.  theta and delta_r come from the GPU residual kernel; its correctly rounded hypot
    and the device atan2 can differ from numpy's by an ulp, which '%.6f' shows on a line in many thousands at most."""
    import io
    import pandas as pd
    from conftest import year_fixture
    from pointcloud_amd.utils.year_export import export_year_data, match_year_slices
    g, results = year_fixture()
    bpath, spath = export_year_data(results, " 2023 ", 2.0, str(tmp_path))
    b2, _ = export_year_data(results[:3], "2021", 1.5, str(tmp_path))
    assert os.path.basename(bpath) == "year_2023_boundary_data.csv" and os.path.basename(spath) == "year_2023_summary.csv"
    assert open(spath, "rb").read() == g["summary_csv"].tobytes()
    for path, key in ((bpath, "boundary_csv"), (b2, "boundary_csv_2021")):
        got = open(path, "rb").read().splitlines()
        want = g[key].tobytes().splitlines()
        assert len(got) == len(want) and got[0] == want[0]
        assert sum(a != b for a, b in zip(got, want)) <= len(want) // 1000, key
        a = pd.read_csv(path)
        b = pd.read_csv(io.BytesIO(g[key].tobytes()))
        assert a.drop(columns=["theta", "delta_r"]).equals(b.drop(columns=["theta", "delta_r"]))
        assert np.allclose(a[["theta", "delta_r"]].to_numpy(), b[["theta", "delta_r"]].to_numpy(), rtol=0, atol=1.01e-6,
                           equal_nan=True)
    files = [bpath, spath, b2]
    for q, (z, t) in enumerate(g["queries"]):
        hits = []
        for k, f in enumerate(files):
            m = match_year_slices(f, z, t)
            if m is None:
                continue
            hits.append(k)
            assert m[0] == str(g[f"match{q}_{k}_year"])
            assert np.array_equal(m[1].index.to_numpy(), g[f"match{q}_{k}_index"])
            assert np.allclose(m[1]["theta"].to_numpy(), g[f"match{q}_{k}_theta"], rtol=0, atol=1.01e-6, equal_nan=True)
        assert hits == g[f"match{q}_files"].tolist(), q


def test_year_export_files_and_matcher(tmp_path):
    """year_{Y}_boundary_data.csv / year_{Y}_summary.csv of a fresh analysis against the row-by-row restatement of
    gui.py:1388-1459 written with the same pandas call (the restatement itself is pinned to the reference's bytes by
    tests/test_oracle_np.py)."""
    import pandas as pd
    from pointcloud_amd import Config, CylinderAnalyzer
    from pointcloud_amd.synth import make_cylinder
    from pointcloud_amd.utils.year_export import export_year_data, match_year_slices
    pts = make_cylinder(60_000, 12, seed=21, outlier_frac=0.02)
    results = CylinderAnalyzer(Config(min_points_per_slice=500)).analyze_with_progress(pts)
    results.append(dict(results[0], z_center=99.0, boundary_points=[]))          # a slice without boundary points
    bpath, spath = export_year_data(results, " 2023 ", 2.0, str(tmp_path))
    assert bpath.endswith("year_2023_boundary_data.csv") and spath.endswith("year_2023_summary.csv")
    want = pd.DataFrame(R.year_boundary_rows(results, "2023", 2.0))
    got = pd.read_csv(bpath)
    assert list(got.columns) == list(want.columns) and len(got) == len(want)
    for col in want.columns:
        if col == "year":
            assert (got[col].astype(str) == "2023").all()
        else:
            # '%.6f' text of values that agree to an ulp: equal up to one unit of the last printed digit
            assert np.allclose(got[col].to_numpy(float), want[col].to_numpy(float), rtol=0, atol=1.01e-6,
                               equal_nan=True), col
    ref_text = want.to_csv(index=False, float_format="%.6f").splitlines()
    got_text = open(bpath).read().splitlines()
    assert sum(a != b for a, b in zip(ref_text, got_text)) <= len(ref_text) // 1000
    summ = pd.read_csv(spath)
    assert list(summ.columns) == ["year", "z_center", "cx", "cy", "R", "thickness", "ovality_pct", "n_boundary_points"]
    assert summ["n_boundary_points"].tolist() == [len(r["boundary_points"]) for r in results]
    year, rows = match_year_slices(bpath, results[3]["z_center"] + 4e-4, 2.0)
    assert year == "2023" and len(rows) == len(results[3]["boundary_points"])
    assert np.all(np.diff(rows["theta"].to_numpy()) >= 0)
    assert match_year_slices(bpath, results[3]["z_center"] + 0.5, 2.0) is None
    assert match_year_slices(pd.DataFrame({"z_center": [1.0]}), 1.0, 2.0) is None


def test_resident_scan_gives_the_same_analysis():
    """A DeviceScan (uploaded once) carries numpy's z range and analyses like the array it came from."""
    from pointcloud_amd import Config, CylinderAnalyzer, DeviceScan
    from pointcloud_amd.synth import make_cylinder
    pts = make_cylinder(200_000, 25, seed=31, outlier_frac=0.03)
    az = CylinderAnalyzer(Config(min_points_per_slice=500))
    want = az.analyze_with_progress(pts)
    with DeviceScan(pts) as scan:
        assert len(scan) == len(pts)
        assert scan.z_min == pts[:, 2].min() and scan.z_max == pts[:, 2].max()
        assert np.array_equal(az.calculate_z_centers(scan), az.calculate_z_centers(pts))
        seen = []
        got = az.analyze_with_progress(scan, progress_callback=seen.append)
        again = az.analyze_with_progress(scan)                  # the copy on the device is reusable
        # single-slice inspection on the resident copy (gui.py:1478-1501)
        zc = az.calculate_z_centers(pts)[7]
        (res_a, cand_a, slc_a, fit_a, ov_a), edge_a, _ = az.process_slice(pts, zc)
        (res_b, cand_b, slc_b, fit_b, ov_b), edge_b, _ = az.process_slice(scan, zc)
        assert res_a == res_b and fit_a == fit_b and np.array_equal(slc_a, slc_b) and edge_a.equals(edge_b)
        assert np.array_equal(ov_a["dev_uniform"], ov_b["dev_uniform"], equal_nan=True)
        # window populations from the count pass alone (closed windows: a point on a shared edge counts twice)
        zcs = az.calculate_z_centers(pts)
        z = pts[:, 2]
        assert np.array_equal(scan.window_counts(zcs - 1.0, zcs + 1.0),
                              [np.count_nonzero((z >= c - 1.0) & (z <= c + 1.0)) for c in zcs])
        assert scan.window_counts([], []).size == 0
    assert seen[-1] == 100 and seen == sorted(seen)
    assert len(got) == len(want) == len(again) > 20
    for a, b, c in zip(want, got, again):
        assert a == b == c
    scan.close()                                                 # closing twice is harmless
    with pytest.raises(F.PcfitError, match="closed"):
        az.analyze_with_progress(scan)
    bad = pts.copy()
    bad[1234, 2] = np.nan
    with DeviceScan(bad) as scan:
        assert np.isnan(scan.z_min) and np.isnan(scan.z_max)
    with pytest.raises(ValueError):
        CylinderAnalyzer(Config(z_min=1.0, z_max=9.0)).analyze_with_progress(DeviceScan(pts))


def test_batch_of_scans_in_one_process():
    """analyze_scans_sharded without a process group: the caller takes every scan, arrays or loaders alike, the
    next upload overlapped with the current analysis; results equal the plain loop."""
    from pointcloud_amd import Config, CylinderAnalyzer, sharding
    from pointcloud_amd.synth import make_cylinder
    az = CylinderAnalyzer(Config(min_points_per_slice=500))
    scans = [make_cylinder(60_000, 10, seed=300 + k, outlier_frac=0.02) for k in range(4)]
    seen = []
    got = sharding.analyze_scans_sharded(az, [scans[0], lambda: scans[1], scans[2], lambda: scans[3]],
                                         progress_callback=seen.append)
    assert seen == [25, 50, 75, 100]
    for k in range(4):
        assert got[k] == az.analyze_with_progress(scans[k])
    assert sharding.analyze_scans_sharded(az, []) == []
    with pytest.raises(ValueError, match="Nx3"):
        sharding.analyze_scans_sharded(az, [np.zeros((5, 2))])


def test_hybrid_with_all_points_as_candidates_in_a_large_batch():
    """use_hull_vertices_only=False on 40 slices of 8000 points: 320 000 candidates come back (the results
    then leave the device section by section); three slices are compared with the same call on the slice alone."""
    x, y, off = R.make_slices(40, 8000, seed=23, outlier_frac=0.1)
    h = F.hybrid_batch(x, y, off, max_trials=64, use_hull=False, want_profile=False)
    assert np.all(h["status"] == 0) and h["cand_offsets"][-1] > 300_000
    for s in (0, 17, 39):
        sl = slice(off[s], off[s + 1])
        one = F.hybrid_batch(x[sl], y[sl], None, max_trials=64, use_hull=False, want_profile=False)
        c0, c1 = h["cand_offsets"][s], h["cand_offsets"][s + 1]
        for k in ("cand_x", "cand_y", "mask", "radius", "theta"):
            assert np.array_equal(h[k][c0:c1], one[k]), k
        assert np.array_equal(h["fit"][s], one["fit"][0]) and np.array_equal(h["ov"][s], one["ov"][0])
        assert h["argmax"][s] == one["argmax"][0] and h["argmin"][s] == one["argmin"][0]
        uniq = R.unique_rows(np.stack([x[sl], y[sl]], 1))
        assert np.array_equal(np.stack([one["cand_x"], one["cand_y"]], 1), uniq)


def test_calls_from_several_threads_at_once():
    """The reference enters the numerics from a pool of threads (analyzer.py:203-207) and ctypes drops the GIL
    during a call: concurrent calls into the library (shared block cache, shared sampling-table cache, one
    null stream) must give what the same calls give one after the other."""
    from concurrent.futures import ThreadPoolExecutor
    jobs = []
    for k in range(12):
        x, y, off = R.make_slices(3 + k % 4, 700 + 450 * k, seed=100 + k, outlier_frac=0.05 * (k % 3))
        jobs.append((x, y, off))

    def run(job, order="qhull"):
        x, y, off = job
        h = F.hybrid_batch(x, y, off, max_trials=500, want_profile=False, order=order)
        k, _ = F.kasa_batch(x, y, off)
        return h["fit"], h["mask"], h["cand_x"], k

    serial = [run(j) for j in jobs]
    for _ in range(2):
        with ThreadPoolExecutor(6) as pool:
            parallel = list(pool.map(run, jobs))
        for a, b in zip(serial, parallel):
            for u, v in zip(a, b):
                assert np.array_equal(u, v, equal_nan=True)

    # Six analyzers at once on scans whose hulls hold outliers (so that the candidate order decides the sampled
    # triples and the fits), half of them in Qhull's order, half from the lexicographic minimum: the order is an
    # argument of each call, so every analysis equals its own serial run whatever the others ask for
    from pointcloud_amd import Config, CylinderAnalyzer
    from pointcloud_amd.synth import make_cylinder
    work = []
    for k in range(6):
        pts = make_cylinder(60_000, 12, seed=300 + k, outlier_frac=0.05)
        order = ("qhull", "canonical")[k % 2]
        work.append((CylinderAnalyzer(Config(z_step=2.0, min_points_per_slice=500, hull_order=order)), pts))
    alone = [az.analyze_with_progress(pts) for az, pts in work]
    for _ in range(3):
        with ThreadPoolExecutor(6) as pool:
            together = list(pool.map(lambda w: w[0].analyze_with_progress(w[1]), work))
        assert together == alone
    # (the two orders do differ on this data: the test would not notice a shared switch otherwise)
    other = [CylinderAnalyzer(Config(z_step=2.0, min_points_per_slice=500,
                                     hull_order=("canonical", "qhull")[k % 2])).analyze_with_progress(work[k][1])
             for k in range(6)]
    assert sum(a != b for a, b in zip(alone, other)) >= 3
    jobs_mixed = [(j, ("qhull", "canonical")[k % 2]) for k, j in enumerate(jobs)]
    serial = [run(j, o) for j, o in jobs_mixed]
    with ThreadPoolExecutor(6) as pool:
        parallel = list(pool.map(lambda a: run(*a), jobs_mixed))
    for a, b in zip(serial, parallel):
        for u, v in zip(a, b):
            assert np.array_equal(u, v, equal_nan=True)


def test_device_entry_points_can_be_captured_in_a_hip_graph():
    """The _dev calls only enqueue (no allocation, no synchronisation): a serving loop may capture one into a
    HIP graph and replay it.  The fused fit + residual launch is captured on a side stream, replayed on fresh
    inputs and compared with direct calls."""
    import ctypes as C
    import torch
    dev = torch.device("cuda", 0)
    S, n, H = 300, 4096, 64
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    p = lambda a: C.c_void_p(a.data_ptr())
    x0, y0, off = R.make_slices(S, n, seed=41)
    x1, y1, _ = R.make_slices(S, n, seed=42, outlier_frac=0.1)
    dx, dy, doff, dedge = t(x0), t(y0), t(off), t(np.linspace(-0.5, 0.5, H + 1))
    kasa, fit = torch.empty(S, 3, dtype=torch.float64, device=dev), torch.empty(S, 3, dtype=torch.float64, device=dev)
    it, st_ = torch.empty(S, 4, dtype=torch.int32, device=dev), torch.empty(S, dtype=torch.int32, device=dev)
    hist = torch.zeros(S, H, dtype=torch.int64, device=dev)
    rmm, amm = torch.empty(S, 2, dtype=torch.float64, device=dev), torch.empty(S, 2, dtype=torch.int64, device=dev)
    side = torch.cuda.Stream()

    def launch(stream):
        F.check(F.lib().pcfit_fit_residuals_batch_dev(p(dx), p(dy), p(doff), S, n, 1.0, 100, 1e-2, 1e-10, 1e-12, p(kasa),
                                                      p(fit), p(it), p(st_), H, p(dedge), p(hist), p(rmm), p(amm),
                                                      C.c_void_p(stream.cuda_stream)))

    launch(side)                                         # (first call outside the capture: one-time function attributes)
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        launch(side)
    for xs, ys in ((x1, y1), (x0, y0)):
        dx.copy_(t(xs))
        dy.copy_(t(ys))
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        got = [a.clone() for a in (kasa, fit, it, st_, hist, rmm, amm)]
        launch(torch.cuda.current_stream())
        torch.cuda.synchronize()
        for a, b in zip(got, (kasa, fit, it, st_, hist, rmm, amm)):
            assert torch.equal(a, b) or (a.is_floating_point() and torch.allclose(a, b, rtol=0, atol=0, equal_nan=True))


def test_large_transfers_arrive_intact():
    """Copies of 32 MB and more are staged through pinned buffers by several host threads, both ways: six
    million points through one all-embracing window come back as they went in (sizes off the chunk grid)."""
    n = 6_000_011
    g = np.random.default_rng(77)
    pts = np.stack([g.normal(size=n), g.normal(size=n), g.uniform(0.0, 1.0, n)], 1)
    off, x, y, perm = F.bin_slices(pts, [-1.0], [2.0])
    assert off[-1] == n and np.array_equal(perm, np.arange(n))
    assert np.array_equal(x, pts[:, 0]) and np.array_equal(y, pts[:, 1])


def test_analyze_file_writes_the_reference_formats(tmp_path, monkeypatch, capsys):
    """analyze_file: text in (native loader), slice_results.csv and slice_points.csv out; the points file, written
    by the library's formatter, equals byte for byte what pandas writes from the same columns."""
    import pandas as pd
    from pointcloud_amd import Config, CylinderAnalyzer
    pts = R.make_cylinder(120_000, 12, seed=51, outlier_frac=0.02)
    scan = tmp_path / "scan.txt"
    np.savetxt(scan, pts, fmt="%.6f", header="x y z")
    monkeypatch.chdir(tmp_path)
    az = CylinderAnalyzer(Config(min_points_per_slice=500))
    results = az.analyze_file(str(scan))
    out = capsys.readouterr().out
    assert "Z range in data" in out and len(results) >= 10
    loaded = np.loadtxt(scan)
    again = az.analyze_with_progress(loaded)
    assert [r["z_center"] for r in results] == [r["z_center"] for r in again]
    assert results[3]["boundary_points"] == again[3]["boundary_points"] and results[3]["R"] == again[3]["R"]
    res_csv = pd.read_csv(tmp_path / "slice_results.csv")
    assert len(res_csv) == len(results) and list(res_csv.columns) == list(results[0].keys())
    # the same edge columns through pandas
    frames = []
    for r in results:
        (pack, edge, warn) = az.process_slice(loaded, r["z_center"])
        frames.append(edge)
    want = pd.concat(frames, ignore_index=True).sort_values(["z_center", "theta"]).reset_index(drop=True)
    want_bytes = want.to_csv(index=False, float_format="%.6f").encode()
    assert (tmp_path / "slice_points.csv").read_bytes() == want_bytes


def test_last_accepted_step_taken_without_its_sweep():
    """Where the cost model proves that a candidate is accepted and everything after it rejected, the kernel takes
    the candidate without sweeping over the points again.  Slices of many sizes, radii, noise levels and arcs, from
    the Kasa start and from shifted starts: parameters, iteration and accepted-step counts of the literal loop."""
    gen = np.random.default_rng(2024)
    sets, inits = [], []
    for i in range(240):
        n = int(gen.choice([40, 200, 1000, 4096, 8192, 5000]))
        rad = float(gen.choice([0.5, 3.0, 10.0, 50.0]))
        sig = min(0.1, rad * float(gen.choice([1e-4, 2e-3, 5e-3, 2e-2])))      # within the Huber threshold of 1
        arc = float(gen.choice([2 * np.pi, 2 * np.pi, np.pi, 0.6 * np.pi]))
        th = gen.uniform(0.0, arc, n) + gen.uniform(0, 6.28)
        rr = rad + gen.normal(0.0, sig, n)
        c = gen.normal(0.0, 5.0, 2)
        xy = np.stack([c[0] + rr * np.cos(th), c[1] + rr * np.sin(th)], 1)
        sets.append(xy)
        _, k = N.kasa(xy[:, 0], xy[:, 1])
        if i % 3 == 2:
            k = k + np.array([0.02 * rad, -0.01 * rad, 0.03 * rad])
        inits.append(k)
    x, y, off = csr(sets)
    inits = np.array(inits)
    fit, iters = F.lm_batch(x, y, off, inits)
    modelled = iters[:, 1] > iters[:, 3]           # an accepted step more than candidates swept
    assert modelled.mean() > 0.3
    same_counts = 0
    for s, xy in enumerate(sets):
        want, it, acc = N.lm(xy[:, 0], xy[:, 1], inits[s])
        assert rel(fit[s], want) < 1e-9, (s, modelled[s], fit[s], want)
        same_counts += (iters[s, 0], iters[s, 1]) == (it, acc)
    assert same_counts >= 0.97 * len(sets)


def test_hull_sort_and_chain_paths_and_their_fall_backs():
    """Slices held in LDS, a few hundred to a few thousand points: rings (counting sort on x, both chains by pruning
    rounds), coordinates on a coarse grid and a vertical bar (buckets too full / all x equal: the sorting network),
    a cascade in which each pruning round can take one candidate only (more rounds than are tried: the chains are
    built one by one), few distinct points, and collinear points.  Distinct points and hull as the CPU gives them."""
    gen = np.random.default_rng(77)
    sets = []
    for n in (300, 1784, 5000, 8192):
        th = gen.uniform(0, 2 * np.pi, n)
        rr = 10.0 + gen.normal(0, 0.05, n)
        sets.append(np.stack([3 + rr * np.cos(th), -2 + rr * np.sin(th)], 1))
    grid = np.round(sets[2] * 2.0) / 2.0                                   # 40 x 40 distinct positions
    bar = np.stack([np.full(700, 1.25), gen.normal(0, 1, 700)], 1)
    bar[::7, 0] = 1.5
    k = np.arange(400.0)
    cascade = np.concatenate([np.stack([k, k * k], 1), [[400.0, -1e9]], [[200.0, 5e5]]])
    few = sets[0][gen.integers(0, 5, 600)]
    line = np.stack([k, 2.0 * k + 1.0], 1)
    sets += [grid, bar, cascade, few, line]
    x, y, off = csr(sets)
    h = F.unique_hull_batch(x, y, off)
    for s, xy in enumerate(sets):
        uniq = R.unique_rows(xy)
        o, nu, nh = off[s], int(h["n_unique"][s]), int(h["n_hull"][s])
        assert nu == len(uniq) and np.array_equal(np.stack([h["ux"][o:o + nu], h["uy"][o:o + nu]], 1), uniq), s
        if s == len(sets) - 1:
            assert h["status"][s] != 0                                      # rank deficient: no hull asked of it
            continue
        assert np.array_equal(h["hull_idx"][o:o + nh], N.hull(uniq)), s
    assert int(h["n_hull"][6]) <= 5                                         # the cascade: next to nothing is left of it


def test_angular_bin_boundary_from_quick_angles_and_radii():
    """The boundary kernel for up to 4096 bins decides bins and maxima from a polynomial angle and the reciprocal-
    square-root radius and runs hypot / atan2 only over a short list of candidates: same selected indices as the
    exact kernel (PCFIT_ANGLE_MAX_EXACT=1) and as the CPU, on points sitting on bin edges and axes, on the centre,
    with the largest radius of a bin attained several times, on a coarse grid, with NaN, for 1 to 4096 bins."""
    gen = np.random.default_rng(5)
    sets = []
    for n in (5, 300, 1023, 1024, 1025, 5000, 40_000):
        th = gen.uniform(0, 2 * np.pi, n)
        rr = 10.0 + gen.normal(0, 0.05, n)
        sets.append(np.stack([3.0 + rr * np.cos(th), -2.0 + rr * np.sin(th)], 1))
    k = np.arange(7200)
    edge = np.stack([3.0 + 10.0 * np.cos(k * (2 * np.pi / 720)), -2.0 + 10.0 * np.sin(k * (2 * np.pi / 720))], 1)
    edge[::3] = edge[1::3][: len(edge[::3])]                                 # radii attained twice
    axes = np.array([[3.0, -2.0], [13.0, -2.0], [-7.0, -2.0], [3.0, 8.0], [3.0, -12.0], [13.0, -2.0], [3.0, -2.0],
                     [13.0, 8.0], [-7.0, 8.0], [-7.0, -12.0], [13.0, -12.0]] * 30)
    grid = np.round(sets[5] * 4.0) / 4.0
    holes = sets[4].copy()
    holes[::7] = np.nan
    holes[1::50, 0] = np.nan
    sets += [edge, axes, grid, holes]
    x, y, off = csr(sets)
    fit = np.tile([3.0, -2.0, 10.0], (len(sets), 1))
    fit[3] = [3.0 + 1e-3, -2.0 - 2e-3, 10.0]
    for bins in (1, 7, 90, 720, 3600, 4096):
        quick = F.angle_max_batch(x, y, off, fit, bins=bins)
        os.environ["PCFIT_ANGLE_MAX_EXACT"] = "1"
        try:
            exact = F.angle_max_batch(x, y, off, fit, bins=bins)
        finally:
            del os.environ["PCFIT_ANGLE_MAX_EXACT"]
        assert np.array_equal(quick, exact), bins
        # (the CPU restatement takes no NaN; with points on the very edges of bins its libm's atan2 and the
        # device's, one ulp apart, may name different bins)
        for s in (0, 2, 6, 8, 9):
            assert np.array_equal(N.angle_max(sets[s], fit[s][0], fit[s][1], bins), quick[s]), (bins, s)
