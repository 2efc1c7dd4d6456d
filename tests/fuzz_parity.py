"""Randomised parity sweep of the GPU entry points against the C restatement (by hand on a GPU box:
``python tests/fuzz_parity.py [rounds] [first_seed]``; tests/test_randomised.py runs a fixed-seed slice of it
under pytest).  Every round draws a batch of
slices whose lengths sit on and around the kernels' shape boundaries and whose content runs from clean rings to
degenerate sets, and compares stage by stage.  Prints one line per mismatch and a summary; exit code 1 on any."""
import os
import sys
import time

import numpy as np
import torch  # noqa: F401  (first, so that the process runs on ONE HIP runtime: the one torch ships)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import native as N           # noqa: E402
from oracle import qhull2d as Q
from oracle import ref_np as R           # noqa: E402
from pointcloud_amd import _ffi as F     # noqa: E402

EDGE_SIZES = [0, 1, 2, 3, 4, 5, 7, 63, 64, 65, 127, 255, 256, 257, 1023, 1024, 1025, 2047, 2048, 2049, 4095, 4096, 4097,
              8191, 8192, 8193, 9000, 16384, 20001]
bad = []


def note(seed, s, what, detail=""):
    bad.append((seed, s, what))
    print(f"MISMATCH seed {seed} slice {s}: {what} {detail}", flush=True)


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))) if a.size else 0.0


def fused_dev(x, y, off, H, lo, hi, max_iter):
    """pcfit_fit_residuals_batch_dev on device copies of the batch."""
    import ctypes as C
    S = len(off) - 1
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    p = lambda a: C.c_void_p(a.data_ptr())
    dx, dy, doff, dedge = t(x if len(x) else np.zeros(1)), t(y if len(y) else np.zeros(1)), t(off), t(np.linspace(lo, hi, H + 1))
    ok_, of_ = torch.empty(S, 3, dtype=torch.float64, device=dev), torch.empty(S, 3, dtype=torch.float64, device=dev)
    oi, os_ = torch.empty(S, 4, dtype=torch.int32, device=dev), torch.empty(S, dtype=torch.int32, device=dev)
    oh = torch.zeros(S, H, dtype=torch.int64, device=dev)
    orm, oam = torch.empty(S, 2, dtype=torch.float64, device=dev), torch.empty(S, 2, dtype=torch.int64, device=dev)
    F.check(F.lib().pcfit_fit_residuals_batch_dev(p(dx), p(dy), p(doff), S, int(np.diff(off).max()), 1.0, max_iter, 1e-2,
                                                  1e-10, 1e-12, p(ok_), p(of_), p(oi), p(os_), H, p(dedge), p(oh), p(orm),
                                                  p(oam), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    c = lambda a: a.cpu().numpy()
    return dict(kasa=c(ok_), fit=c(of_), iters=c(oi), status=c(os_), hist=c(oh), rmm=c(orm), amm=c(oam))


def make_slice(g, n):
    kind = g.choice(["ring", "ring", "ring", "outliers", "arc", "dups", "line", "same", "tiny", "far", "grid"])
    th = g.uniform(0, 2 * np.pi, n)
    r = 10.0 + g.normal(0, 0.05, n)
    c = np.array([3.0, -2.0])
    if kind == "outliers":
        m = g.random(n) < g.uniform(0.02, 0.4)
        r[m] = g.uniform(0, 14.0, int(m.sum()))
    if kind == "arc":
        th = g.uniform(0.3, 0.3 + g.uniform(0.2, 3.0), n)
    xy = np.stack([c[0] + r * np.cos(th), c[1] + r * np.sin(th)], 1)
    if kind == "dups" and n > 4:
        xy[g.integers(0, n, n // 3)] = xy[g.integers(0, n, n // 3)]
    if kind == "line":
        t = g.normal(0, 5, n)
        xy = np.stack([1.0 + 2.0 * t, -3.0 + 0.5 * t], 1)
    if kind == "same":
        xy = np.tile([[1.25, -7.5]], (n, 1))
    if kind == "tiny":
        xy = (xy - c) * 1e-6
    if kind == "far":
        xy = xy + np.array([1.0e6, -2.0e6])
    if kind == "grid":
        xy = np.round(xy * 8.0) / 8.0                    # many exact duplicates and collinear runs
    return kind, np.ascontiguousarray(xy)


def few_points_tol(xy, radius):
    """What three or four points leave to rounding, against the circle's size: 1e-7 when they are almost in line (a
    circle twenty times their spread: one coordinate moved by a unit in the last place moves it by 4e-9), and 1e-9
    times a thousandth of (spread / least distance between two of them)^2 when two of them nearly coincide - the
    normal equations square that ratio (points 0.03 apart and 28 from the third: two kernels 1.1e-9 apart; 6.5e-8 apart and 2.4e-5 from the third: 3.0e-8)."""
    spread = float(np.ptp(xy[:, 0]) + np.ptp(xy[:, 1]))
    if abs(radius) > 20.0 * spread:
        return 1e-7
    u = np.unique(xy, axis=0)
    d = min(float(np.hypot(*(u[i] - u[j]))) for i in range(len(u)) for j in range(i))
    return 1e-9 * max(1.0, 1e-3 * (spread / d) ** 2) if d > 0 else 1e-7


def in_one_line(xy):
    """Fewer than three distinct points, or all of them exactly in one line (a handful of points on a coarse grid can
    be): no circle is determined, and what a fit returns is the rounding of a zero determinant - on both sides."""
    u = np.unique(xy, axis=0)
    if len(u) < 3:
        return True
    d = u[1:] - u[0]
    return bool(np.all(d[0, 0] * d[1:, 1] - d[0, 1] * d[1:, 0] == 0.0))


def lm_on_a_knife_edge(xy, init, got, tol, scale):
    """Is `got` what the C loop returns when its threshold for accepting a step (cost < cost_prev - 1e-12) is moved
    by about its own size?  On a short arc, or with a handful of points and outliers among them, the loop's last
    accept / reject decisions are left to the rounding of its cost sums: two such slices in two million took one accepted step more in the kernels than in the C loop (13
    against 12, results 1.6e-6 and 2.0e-6 apart), and so does the C loop itself once fifty coordinates move by one
    unit in the last place."""
    runs = []
    for tol_cost in (0.0, 0.5e-12, 1e-12, 2e-12, 4e-12):
        w2, _, _ = N.lm(np.ascontiguousarray(xy[:, 0]), np.ascontiguousarray(xy[:, 1]), init, tol_cost=tol_cost)
        err = float(np.max(np.abs(got - w2)) / scale) if scale else rel(got, w2)
        if err <= tol:
            return True
        runs.append(w2)
    # A loop that crawls along a flat valley (seven points, outliers among them: 54 accepted steps of 1e-12 each
    # in the C loop, 55 in the kernels) ends wherever the threshold stops it: the kernels' circle then lies on the
    # loop's path between the circles of two neighbouring thresholds.
    size = np.abs(runs[2]).max()
    for lo, hi in zip(runs[:-1], runs[1:]):
        if np.all(got >= np.minimum(lo, hi) - 1e-9 * size) and np.all(got <= np.maximum(lo, hi) + 1e-9 * size):
            return True
    return False


def one_round(seed):
    g = np.random.default_rng(seed)
    S = int(g.integers(3, 14))
    sizes = [int(g.choice(EDGE_SIZES)) if g.random() < 0.6 else int(g.integers(0, 9000)) for _ in range(S)]
    kinds, sets = zip(*(make_slice(g, n) for n in sizes))
    x = np.concatenate([s[:, 0] for s in sets]) if sum(sizes) else np.zeros(0)
    y = np.concatenate([s[:, 1] for s in sets]) if sum(sizes) else np.zeros(0)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)

    kasa, kst = F.kasa_batch(x, y, off)
    init = np.where(np.isfinite(kasa), kasa, 1.0)
    lm, iters = F.lm_batch(x, y, off, init)
    res = F.residuals_batch(x, y, off, init, bins=64, lo=-0.5, hi=0.5)
    # statistics only: another kernel (the fused kernel's residual pass about the given circles)
    stat = F.residuals_batch(x, y, off, init, bins=64, lo=-0.5, hi=0.5, want_theta=False, want_dr=False)
    for key in ("hist", "rmax", "rmin", "argmax", "argmin"):
        if not np.array_equal(stat[key], res[key], equal_nan=True):
            differ = np.flatnonzero([not np.array_equal(stat[key][s], res[key][s], equal_nan=True) for s in range(S)])
            note(seed, int(differ[0]), "statistics-only residual pass", f"{key} [{kinds[differ[0]]} n={sizes[differ[0]]}]")
    # the one-launch kernels (Kasa + LM + residual statistics; Kasa + residual statistics), device entry point
    for max_iter in (100, 0):
        fu = fused_dev(x, y, off, 64, -0.5, 0.5, max_iter)
        okf = np.isfinite(fu["fit"]).all(axis=1)
        # (exactly collinear or identical points: whether the determinant rounds to zero depends on the summation order)
        plain = np.array([k not in ("line", "same") and not (len(v) <= 16 and in_one_line(v)) for k, v in zip(kinds, sets)])
        if not np.array_equal(fu["status"][plain], kst[plain]):
            note(seed, -1, "fused status", f"max_iter {max_iter}")
        about = np.where(okf[:, None], fu["fit"], 1.0)
        chk = F.residuals_batch(x, y, off, about, bins=64, lo=-0.5, hi=0.5)          # the per-point kernel
        for s in np.flatnonzero(okf):
            tag = f"[{kinds[s]} n={sizes[s]} max_iter={max_iter}]"
            if not (np.array_equal(fu["hist"][s], chk["hist"][s]) and fu["amm"][s, 0] == chk["argmax"][s]
                    and fu["amm"][s, 1] == chk["argmin"][s] and fu["rmm"][s, 0] == chk["rmax"][s]
                    and fu["rmm"][s, 1] == chk["rmin"][s]):
                note(seed, int(s), "fused residual statistics", tag)
            if max_iter == 0 and not np.array_equal(fu["fit"][s], fu["kasa"][s]):
                note(seed, int(s), "fused fit without LM is not its Kasa circle", tag)
            if kinds[s] in ("ring", "dups", "tiny", "grid") and sizes[s] >= 3 and not (sizes[s] <= 16 and in_one_line(sets[s])):
                want = lm[s] if max_iter else kasa[s]
                err = float(np.max(np.abs(fu["fit"][s] - want)) / np.abs(want).max())     # against the circle's size
                if not err <= (few_points_tol(sets[s], want[2]) if sizes[s] <= 4 else 1e-9):
                    note(seed, int(s), "fused fit", f"{tag} {err:.2e}")
    sel = F.angle_max_batch(x, y, off, init, bins=90)
    uh = F.unique_hull_batch(x, y, off)
    rs = F.ransac_batch(x, y, off, max_trials=150)
    for s, xy in enumerate(sets):
        n = len(xy)
        tag = f"[{kinds[s]} n={n}]"
        o0, o1 = off[s], off[s + 1]
        # Kasa
        st, k = N.kasa(xy[:, 0], xy[:, 1]) if n else (1, np.full(3, np.nan))
        if n >= 3:
            # (exactly collinear points: whether the determinant rounds to zero depends on the summation order)
            if st != kst[s] and kinds[s] not in ("line", "same") and not (n <= 16 and in_one_line(xy)):
                note(seed, s, "kasa status", f"{tag} {kst[s]} vs {st}")
            # (a million units off the origin a three- or four-point fit loses up to 1e-8: conditioning, both sides)
            # errors are measured against the circle's size: a centre coordinate may be arbitrarily close to zero
            elif st == 0 and kinds[s] not in ("line", "same") and not (n <= 16 and in_one_line(xy)):
                err = float(np.max(np.abs(kasa[s] - k)) / np.abs(k).max())
                if not err <= (1e-6 if kinds[s] == "far" else few_points_tol(xy, k[2]) if n <= 4 else 1e-9):
                    note(seed, s, "kasa value", f"{tag} {err:.2e}")
        # LM from the same start
        if n >= 3 and kinds[s] not in ("line", "same") and not (n <= 16 and in_one_line(xy)):
            want, it, acc = N.lm(xy[:, 0], xy[:, 1], init[s])
            tol = 1e-9 if kinds[s] in ("ring", "dups", "tiny", "grid") else 1e-6
            # (a short noisy arc fits a near-line: the radius runs off to hundreds of times the true one and a
            # 1e-15 perturbation of the input moves the CPU result by 1e-6 - conditioning, not arithmetic)
            if kinds[s] == "arc" and abs(want[2]) > 100.0:
                tol = 1e-4
            # (the same for a handful of points almost in line, whatever their kind: a loop that runs off to a circle a
            # hundred times their spread - radius 28 046 after 94 accepted steps, still growing - has no limit to agree on)
            if n <= 16 and abs(want[2]) > 100.0 * float(np.ptp(xy[:, 0]) + np.ptp(xy[:, 1])):
                tol = 1e-4
            # (errors against the circle's size where a centre coordinate may be arbitrarily close to zero)
            scale = np.abs(want).max() if kinds[s] in ("far", "tiny") else None
            err = float(np.max(np.abs(lm[s] - want)) / scale) if scale else rel(lm[s], want)
            if not (err <= tol) and not lm_on_a_knife_edge(xy, init[s], lm[s], tol, scale):
                note(seed, s, "lm value", f"{tag} {err:.2e}")
        # residual statistics about the start circle
        if n >= 1:
            want = N.residuals(xy, *init[s], bins=64, lo=-0.5, hi=0.5)
            if not np.array_equal(res["hist"][s], want["hist"]):
                note(seed, s, "histogram", tag)
            if res["argmax"][s] != want["argmax"] or res["argmin"][s] != want["argmin"]:
                note(seed, s, "argmax/argmin", tag)
            if res["rmax"][s] != want["rmax"] or res["rmin"][s] != want["rmin"]:
                note(seed, s, "rmax/rmin", tag)
            if not np.array_equal(res["dr"][o0:o1], want["dr"]):
                note(seed, s, "delta r", tag)
            if np.max(np.abs(res["theta"][o0:o1] - want["theta"])) > 2e-15:
                note(seed, s, "theta", tag)
            want_sel = N.angle_max(xy, init[s][0], init[s][1], bins=90)
            if not np.array_equal(sel[s], want_sel):
                # a point within an ulp of a bin edge may land on either side (atan2 of host and device)
                th = want["theta"]
                near = np.min(np.abs((th / (2 * np.pi / 90)) - np.round(th / (2 * np.pi / 90)))) < 1e-9
                if not near:
                    note(seed, s, "angle_max", tag)
        # unique + hull
        if n >= 1:
            uniq = N.sort_unique(xy)
            nu, nh = int(uh["n_unique"][s]), int(uh["n_hull"][s])
            got_u = np.stack([uh["ux"][o0:o0 + nu], uh["uy"][o0:o0 + nu]], 1)
            if len(uniq) >= 3 and N.has_2d_variation(uniq) and uh["status"][s] == 0:
                hv = uniq[N.hull(uniq)]
                if not np.array_equal(got_u[uh["hull_idx"][o0:o0 + nh]], hv):
                    note(seed, s, "hull", tag)
                if not np.array_equal(got_u, uniq):
                    note(seed, s, "unique", tag)
            elif len(uniq) >= 3 and N.has_2d_variation(uniq) != (uh["status"][s] == 0) and kinds[s] not in ("line",):
                note(seed, s, "rank check", f"{tag} status {uh['status'][s]}")
        # RANSAC
        if n >= 3:
            want = N.ransac(xy, trials=150)
            ok_gpu = rs["status"][s] == 0
            if (want["status"] == 0) != ok_gpu:
                note(seed, s, "ransac status", f"{tag} {rs['status'][s]} vs {want['status']}")
            elif ok_gpu:
                if rs["score"][s] != want["score"] or rs["best_trial"][s] != want["best_trial"]:
                    note(seed, s, "ransac score/trial", f"{tag} {rs['score'][s]}/{rs['best_trial'][s]} vs "
                                                        f"{want['score']}/{want['best_trial']}")
                elif not np.array_equal(rs["mask"][o0:o1], want["mask"]):
                    note(seed, s, "ransac mask", tag)
    return S, sum(sizes)


def hybrid_round(seed):
    """The whole hybrid chain (unique, rank check, hull, RANSAC, Kasa, LM, metrics) against the numpy
    restatement, with hull candidates and with all distinct points as candidates."""
    g = np.random.default_rng(seed)
    S = int(g.integers(2, 6))
    sizes = [int(g.choice([0, 2, 3, 5, 40, 300, 1024, 1025, 2500, 6000, 8192])) for _ in range(S)]
    kinds, sets = zip(*(make_slice(g, n) for n in sizes))
    x = np.concatenate([s[:, 0] for s in sets]) if sum(sizes) else np.zeros(0)
    y = np.concatenate([s[:, 1] for s in sets]) if sum(sizes) else np.zeros(0)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    use_hull = bool(g.random() < 0.7)
    trials = int(g.choice([50, 200]))
    # (without the list of distinct points the hull kernel drops interior points before its sort)
    want_unique = bool(g.random() < 0.4)
    # hulls walked from their lexicographic minimum, or as scipy's ConvexHull walks them (the replay of Qhull's
    # construction: over the sorted distinct points when they are wanted, over the slices as they are otherwise)
    qhull = bool(use_hull and g.random() < 0.5)
    h = F.hybrid_batch(x, y, off, max_trials=trials, use_hull=use_hull, want_profile=False, want_unique=want_unique,
                       order="qhull" if qhull else "canonical")
    for s, xy in enumerate(sets):
        tag = f"[{kinds[s]} n={len(xy)} hull={use_hull} unique={want_unique} qhull={qhull}]"
        if use_hull is False and len(xy) > 1024:
            continue                                      # (the restatement's RANSAC loop is too slow for these)
        ref = R.hybrid_fit(np.column_stack([xy, np.zeros(len(xy))]), 0.0, 1.0, max_trials=trials,
                           use_hull_vertices_only=use_hull, hull_start="qhull" if qhull else None) if len(xy) else None
        ok = h["status"][s] == 0
        if (ref is not None) != ok:
            if kinds[s] not in ("line", "same"):
                note(seed, s, "hybrid status", f"{tag} {h['status'][s]} vs {'fit' if ref else 'None'}")
            continue
        if not ok:
            continue
        c0, c1 = h["cand_offsets"][s], h["cand_offsets"][s + 1]
        if not np.array_equal(np.stack([h["cand_x"][c0:c1], h["cand_y"][c0:c1]], 1), ref["candidates_xy"]):
            # (no allowance for slices the replay is not sure of - points exactly in line - either: the
            # restatement equals scipy's list on every such slice tried, and the library has to equal it)
            note(seed, s, "hybrid candidates", tag)
        elif not np.array_equal(h["mask"][c0:c1], ref["inlier_mask"]):
            note(seed, s, "hybrid inlier mask", tag)
        else:
            want = np.array([ref["center_x"], ref["center_y"], ref["radius"]])
            scale = np.abs(want).max() if kinds[s] == "far" else None
            err = float(np.max(np.abs(h["fit"][s] - want)) / scale) if scale else rel(h["fit"][s], want)
            if not err <= (1e-4 if kinds[s] == "arc" and abs(want[2]) > 100.0 else 1e-6):   # (near-line, as above)
                # The reference's last accept / reject decisions can hang on the rounding of its cost sums (short
                # arcs, heavy outliers): then the plain C loop leaves numpy's result by as much, and it is the C
                # loop, operation for operation, that the kernels are held to.
                inl = ref["candidates_xy"][ref["inlier_mask"]]
                _, k0 = N.kasa(np.ascontiguousarray(inl[:, 0]), np.ascontiguousarray(inl[:, 1]))
                c_fit, _, _ = N.lm(np.ascontiguousarray(inl[:, 0]), np.ascontiguousarray(inl[:, 1]), k0)
                if not rel(h["fit"][s], c_fit) <= 1e-9:
                    note(seed, s, "hybrid fit", f"{tag} {err:.2e} (C loop: {rel(h['fit'][s], c_fit):.2e})")
    return S, sum(sizes)


def bin_round(seed):
    """Slice binning: window counts from 1 to a few thousand, shared or separated edges, z continuous or on a
    grid that hits the edges, NaN and out-of-range values."""
    g = np.random.default_rng(seed)
    n = int(g.choice([0, 1, 63, 64, 65, 4095, 4096, 4097, 100_000, 300_000]))
    S = int(g.choice([1, 2, 3, 31, 32, 33, 57, 200, 255, 256, 257, 511, 512, 513, 1000, 3000]))
    step = float(g.choice([2.0, 0.5, 0.1, 1.0 / 3.0]))
    width = step * float(g.choice([1.0, 1.0, 0.5, 0.999]))
    z0 = float(g.choice([0.0, -17.25, 1000.0]))
    lo = z0 + step * np.arange(S)
    hi = lo + width
    z = g.uniform(z0 - 2 * step, z0 + (S + 2) * step, n)
    mode = g.choice(["cont", "grid", "edges"])
    if mode == "grid":
        z = np.round(z / (step / 4)) * (step / 4)
    if mode == "edges" and n:
        k = g.integers(0, S, n)
        z = np.where(g.random(n) < 0.5, lo[k], hi[k])
        z = np.where(g.random(n) < 0.2, np.nextafter(z, g.choice([-np.inf, np.inf])), z)
    if n > 10:
        z[g.integers(0, n, 3)] = np.nan
    pts = np.stack([g.normal(size=n), g.normal(size=n), z], 1) if n else np.zeros((0, 3))
    # more than 256 windows: in one pass, or through groups of 256 windows
    os.environ["PCFIT_BIN_TWO_LEVEL_MIN"] = str(g.choice([-1, 0]))
    off, x, y, perm = F.bin_slices(pts, lo, hi)
    del os.environ["PCFIT_BIN_TWO_LEVEL_MIN"]
    o_ref, p_ref = N.bin_slices(pts, lo, hi)
    if not np.array_equal(off, o_ref):
        note(seed, -1, "bin offsets", f"[n={n} S={S} step={step} width={width} {mode}]")
    elif not np.array_equal(perm, p_ref):
        note(seed, -1, "bin order", f"[n={n} S={S} step={step} width={width} {mode}]")
    elif not (np.array_equal(x, pts[perm, 0]) and np.array_equal(y, pts[perm, 1])):
        note(seed, -1, "bin payload", f"[n={n} S={S}]")
    return S, n


def sweep(rounds, first):
    """``rounds`` rounds of every kind from seed ``first`` on: (slices, points); mismatches are in ``bad``."""
    t0 = time.perf_counter()
    slices = points = 0
    for k in range(rounds):
        s, p = one_round(first + k)
        slices += s
        points += p
        s, p = bin_round(first + k)
        slices += s
        points += p
        s, p = hybrid_round(first + k)
        slices += s
        points += p
        if (k + 1) % 10 == 0:
            print(f"round {k + 1}/{rounds}: {slices} slices, {points} points, {len(bad)} mismatches, "
                  f"{time.perf_counter() - t0:.0f} s", flush=True)
    return slices, points


if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    slices, points = sweep(rounds, first)
    print(f"done: {rounds} rounds, {slices} slices, {points} points, {len(bad)} mismatches")
    sys.exit(1 if bad else 0)
