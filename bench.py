#!/usr/bin/env python3
"""Throughput of the fit + residual pass on MI355X.

One step = one pass of the hot path over one batch of pre-binned slices that is
already resident in HBM, in ONE launch (pcfit_fit_residuals_batch_dev): Kasa
start, LM/Huber refinement and the residual pass (delta-r histogram of 256 bins
over +-10 sigma, Rmax / Rmin and their indices) with each slice read from HBM
once and held on chip (half in registers, half in LDS) for all passes.
Workload at N=1: 10 000 slices x 8192 points, FP64, synthetic noisy cylinder
(the "1x MI355X batched algebraic+geometric fit" configuration).  With N > 1
every rank owns its own batch of that size (slices shard with no data exchange;
weak scaling) and the fitted parameters and histograms of all ranks are
all-gathered with RCCL, overlapped with the next step's kernels.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from pointcloud_amd import _ffi as F            # noqa: E402
from pointcloud_amd.synth import make_slices    # noqa: E402

# How the flop figure of the kernel is counted: the arithmetic of the short form of each pass as the kernel
# executes it, per point, a fused multiply-add as 2, everything else (add, mul, compare, min / max, floor, the
# hardware reciprocal square root) as 1; integer work, moves, lane exchanges and the reductions are not counted.
#   Kasa power sums: u, v (2 sub), u^2 (mul), z = u^2 + v^2 (fma), 4 adds, 3 fma                      = 15
#   first LM sweep: distance (2 sub, mul, fma), rsqrt + third-order step (9), ri (fma), residual (sub),
#             Jacobian (2 mul), 6 fma + 3 add into H, g, cost, curvature, largest / smallest distance (2) = 36
#   candidate sweeps: the same + the three sums of the second-order model (2 mul, 2 fma, add)          = 43
#   residual pass: distance (5), rsqrt + step (9), radius (mul), two tests for the extremes, bin
#             (fma, floor, 2 sub, compare)                                                            = 23
# The reference's literal loop evaluates residuals and Jacobian separately in each of its 100 iterations
# (~1e4 flop per point); the kernel reaches the same accept/reject sequence in 2 sweeps (a third for a slice
# in fifty).  The vector units issue ~132 instructions per point for these ~118 flop.
FLOP_KASA = 15.0
FLOP_SWEEP_FIRST = 36.0
FLOP_SWEEP = 43.0
FLOP_RESID = 23.0
PEAK_FP64_VALU_TFLOPS = 78.6
SETTLE = 64          # launches before the warmup steps: the clock of an idle GPU needs them
RING = 8                      # record buffers in flight between the fit kernel and the all-gather
PEAK_HBM_GBS = 8000.0


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def kernel_sources_sha1():
    """Digest of the sources the fused fit + residual kernel is built from: counters recorded under profiles/ speak
    for the kernel only as long as this has not changed."""
    import hashlib
    h = hashlib.sha1()
    for name in ("fit.hip", "fit_math.h", "pcfit_internal.h", "angle_math.h", "segments.h"):
        with open(os.path.join(ROOT, "pointcloud_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _cpu_step_np(x, y, off, s, bins, lo, hi):
    """The step for slice s as the numpy restatement of the reference evaluates it (Kasa -> LM -> residuals)."""
    from oracle import ref_np as R
    sl = slice(off[s], off[s + 1])
    xs, ys = x[sl], y[sl]
    k = R.kasa_fit(xs, ys)
    f = R.lm_huber(xs, ys, *k)
    theta, r = R.polar(np.stack([xs, ys], 1), f[0], f[1])
    R.residual_histogram(r - f[2], bins, lo, hi)
    int(np.argmax(r)), int(np.argmin(r))
    return xs.size


def cpu_baseline(x, y, off, bins, lo, hi, budget_s=10.0, max_slices=2000):
    """numpy restatement of the same step, slice after slice, on one host thread: the stand-in for the
    reference's serial loop (analyzer.py:148), which cannot travel."""
    t0 = time.perf_counter()
    done = pts = 0
    while done < max_slices and time.perf_counter() - t0 < budget_s:
        pts += _cpu_step_np(x, y, off, done, bins, lo, hi)
        done += 1
    dt = time.perf_counter() - t0
    return pts / dt, done, dt


def cpu_baseline_pool(x, y, off, bins, lo, hi, threads, n_slices):
    """The same slices through ThreadPoolExecutor(threads), the reference's own parallel driver
    (analyzer.py:203-207: one task per slice)."""
    from concurrent.futures import ThreadPoolExecutor
    n_slices = min(n_slices, len(off) - 1)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as pool:
        pts = sum(pool.map(lambda s: _cpu_step_np(x, y, off, s, bins, lo, hi), range(n_slices)))
    dt = time.perf_counter() - t0
    return pts / dt, n_slices, dt


def cpu_baseline_c(x, y, off, bins, lo, hi, n_slices=600, threads=1):
    """Plain C restatement (literal 100-iteration LM loop, no interpreter in the way), on `threads` host
    threads (ctypes releases the GIL during the call), each taking a contiguous share of the slices."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import native as N
    n_slices = min(n_slices, len(off) - 1)
    cuts = np.linspace(0, n_slices, threads + 1).astype(int)

    def share(t):
        a, b = cuts[t], cuts[t + 1]
        if b > a:
            N.fit_residual_batch(x[off[a]:off[b]], y[off[a]:off[b]], off[a:b + 1] - off[a], bins, lo, hi)

    N.lib()
    t0 = time.perf_counter()
    if threads == 1:
        share(0)
    else:
        with ThreadPoolExecutor(max_workers=threads) as pool:
            list(pool.map(share, range(threads)))
    dt = time.perf_counter() - t0
    return off[n_slices] / dt, n_slices, dt


def settle(fn, ms=40.0):
    """Runs fn (GPU work, enqueued) again and again for `ms` of wall clock, at least twice, and waits for it: the clock
    of an idle GPU takes some 20 ms of load to come up, and a timing taken before that is the timing of the ramp."""
    t0 = time.perf_counter()
    k = 0
    while k < 2 or time.perf_counter() - t0 < ms * 1e-3:
        fn()
        torch.cuda.synchronize()
        k += 1


def extra_configs(L, dev, stream):
    """Timings of the other configurations, outside the timed region (events on the kernels' stream):
    seeded RANSAC on 10 000 slices x 8192 points with 30 % outliers and 1024 hypotheses per slice; a 100 M-point
    scan in random z order: binning into 2000 windows (both steps of the two-level form, nothing read back), then
    Kasa + LM fits, the 720-bin boundary and the 256-bin residual histograms of the 2000 windows, and the analyzer's
    hybrid analysis of the same scan kept resident."""
    out = {}

    def timed(fn, reps):
        settle(fn)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    S, n, T = 10_000, 8192, 1024
    g = torch.Generator(device=dev)
    g.manual_seed(2)
    ang = torch.rand(S * n, generator=g, device=dev, dtype=torch.float64) * (2 * np.pi)
    r = 10.0 + 0.05 * torch.randn(S * n, generator=g, device=dev, dtype=torch.float64)
    bad = torch.rand(S * n, generator=g, device=dev, dtype=torch.float64) < 0.3
    r = torch.where(bad, torch.rand(S * n, generator=g, device=dev, dtype=torch.float64) * 14.0, r)
    x, y = 3.0 + r * torch.cos(ang), -2.0 + r * torch.sin(ang)
    del ang, r, bad
    off = torch.arange(S + 1, device=dev, dtype=torch.int64) * n
    tab = torch.from_numpy(F.choice3_table(n, T)).to(dev)
    tof = torch.zeros(S, dtype=torch.int32, device=dev)
    model = torch.empty(S, 3, dtype=torch.float64, device=dev)
    score, best, st = (torch.empty(S, dtype=torch.int32, device=dev) for _ in range(3))
    mask = torch.empty(S * n, dtype=torch.uint8, device=dev)
    ws = torch.empty(L.pcfit_ransac_workspace_bytes(S), dtype=torch.uint8, device=dev)
    ms = timed(lambda: F.check(L.pcfit_ransac_batch_dev(ptr(x), ptr(y), ptr(off), S, 0.5, T, 0.1, None, ptr(tab),
                                                        ptr(tof), ptr(model), ptr(score), ptr(best), ptr(mask), ptr(st),
                                                        ptr(ws), ws.numel(), stream)), 2)
    evals = float(S) * n * T
    out["ransac_10000x8192_1024_hypotheses_30pct_outliers"] = {
        "ms": ms, "points_per_s": S * n / (ms * 1e-3), "point_hypotheses_per_s": evals / (ms * 1e-3),
        "tflops_at_7_flop": evals * 7 / (ms * 1e-3) / 1e12,
        "valu_fp64_frac": evals * 7 / (ms * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS,
        "min_score": int(score.min().item()), "all_ok": bool((st == 0).all().item()),
        "roofline": {"kernel": "ransac_score_kernel (1024 hypotheses a slice, 7 flop a point and hypothesis)",
                     "bound": "valu_fp64", "achieved": evals * 7 / (ms * 1e-3) / 1e12, "peak": PEAK_FP64_VALU_TFLOPS,
                     "unit": "TFLOP/s", "frac": evals * 7 / (ms * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS}}
    del x, y, off, tab, tof, model, score, best, st, mask, ws
    torch.cuda.empty_cache()

    # a 100 M-point scan of a noisy cylinder (sigma 0.05 about R = 10, 10 % outliers anywhere inside 1.4 R), rows in
    # random z order, 2000 windows of ~50 000 points: binning, then fits, angular-bin boundary and histograms
    n, S = 100_000_000, 2000
    xyz = torch.empty(n, 3, device=dev, dtype=torch.float64)
    xyz[:, 2] = torch.rand(n, generator=g, device=dev, dtype=torch.float64)
    xyz[:, 2] *= 2.0 * S
    xyz[:, 0] = torch.rand(n, generator=g, device=dev, dtype=torch.float64)          # angle / 2 pi
    xyz[:, 1] = 10.0 + 0.05 * torch.randn(n, generator=g, device=dev, dtype=torch.float64)
    bad = torch.rand(n, generator=g, device=dev, dtype=torch.float64) < 0.1
    xyz[:, 1] = torch.where(bad, torch.rand(n, generator=g, device=dev, dtype=torch.float64) * 14.0, xyz[:, 1])
    del bad
    ang = xyz[:, 0] * (2 * np.pi)
    rr = xyz[:, 1].clone()
    xyz[:, 0] = 3.0 + rr * torch.cos(ang)
    xyz[:, 1] = -2.0 + rr * torch.sin(ang)
    del ang, rr
    lo = 2.0 * torch.arange(S, device=dev, dtype=torch.float64)
    hi = lo + 2.0
    geom = np.array([0.0, 2.0, 2.0 * (S - 1)])
    wsb = L.pcfit_bin2_workspace_bytes(n, S, 0, n + n // 8)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    off = torch.empty(S + 1, dtype=torch.int64, device=dev)
    status = torch.empty(1, dtype=torch.int32, device=dev)
    xo, yo = torch.empty(n, dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.float64, device=dev)

    def count():
        F.check(L.pcfit_bin2_count_dev(ptr(xyz), n, ptr(lo), ptr(hi), S, F._p(geom), 0, n + n // 8, ptr(off), ptr(status),
                                       ptr(ws), wsb, stream))

    def scatter():
        F.check(L.pcfit_bin2_scatter_dev(n, S, F._p(geom), 0, n + n // 8, ptr(off), ptr(xo), ptr(yo), None, ptr(ws), wsb,
                                         stream))
    ms_count = timed(count, 10)
    ms_scatter = timed(scatter, 10)
    ms = ms_count + ms_scatter
    out["binning_100M_points_2000_windows_random_z_order"] = {
        "ms": ms, "ms_level1_and_counts": ms_count, "ms_level2_scatter": ms_scatter,
        "points_per_s": n / (ms * 1e-3), "status": int(status.item()), "members": int(off[-1].item()),
        "gbs_at_64_bytes_per_point_single_pass": 64.0 * n / (ms * 1e-3) / 1e9,
        "hbm_frac_at_64_bytes_per_point": 64.0 * n / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
        # against what a single pass would move (24 read, 16 written... counted as 64 with the count pass); the two
        # levels move 104.6 bytes a point by the counters (profiles/r03_bin100m_pmc_hbm.txt)
        "roofline": {"kernel": "bin_count + bin_scatter_staged level 1 + level 2 (five kernels and their scans)",
                     "bound": "hbm", "achieved": 64.0 * n / (ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "frac": 64.0 * n / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "algorithmic_bytes": 64.0 * n,
                     "traffic": 104.59 * n, "traffic_source": "profiles/r03_bin100m_pmc_hbm.txt"}}
    # the analyzer's own path on that scan (CylinderAnalyzer.analyze_with_progress underneath: binning, interior
    # filter, unique + hull, seeded RANSAC on the hull, Kasa + LM on the inliers, ovality), the scan resident in
    # HBM (pcfit_scan_upload once, as the reference keeps a loaded scan in memory); per-window results and the
    # hull candidates come back to the host in every call, which is timed by the host clock
    from pointcloud_amd import DeviceScan
    host = xyz.cpu().numpy()
    zc = np.arange(1.0, 2.0 * S - 1.0 + 1e-9, 2.0)
    t_up = time.perf_counter()
    with DeviceScan(host) as scan:
        t_up = time.perf_counter() - t_up
        settle(lambda: F.scan_hybrid_batch(scan, zc - 1.0, zc + 1.0, order="canonical"))
        t_an = time.perf_counter()
        for _ in range(5):
            res = F.scan_hybrid_batch(scan, zc - 1.0, zc + 1.0, order="canonical")
        t_an = (time.perf_counter() - t_an) / 5
        # the same with every hull listed as scipy's ConvexHull lists it (the analyzer's default: a kernel replays
        # Qhull's construction per window, the few windows it hands back are replayed on host cores): the
        # reference's candidates element for element
        settle(lambda: F.scan_hybrid_batch(scan, zc - 1.0, zc + 1.0, order="qhull"))
        t_q = time.perf_counter()
        for _ in range(5):
            res_q = F.scan_hybrid_batch(scan, zc - 1.0, zc + 1.0, order="qhull")
        t_q = (time.perf_counter() - t_q) / 5
    t_up2 = time.perf_counter()
    with DeviceScan(host):
        t_up2 = time.perf_counter() - t_up2
    okw = res["status"] == 0
    out["hybrid_analysis_100M_points_2000_windows_resident"] = {
        "ms_per_analysis": t_an * 1e3, "points_per_s": n / t_an, "windows_fitted": int(okw.sum()),
        # (the reference fits the hull's vertices: with outliers beyond the ring that is the outliers' outline)
        "median_radius": float(np.median(res["fit"][okw, 2])) if okw.any() else None,
        "candidates_returned": int(res["cand_offsets"][-1]),
        "ms_per_analysis_in_qhulls_vertex_order": t_q * 1e3,
        "same_candidate_sets_in_qhulls_order": bool(np.array_equal(res["cand_offsets"], res_q["cand_offsets"])),
        "ms_upload_from_pageable_host_memory": t_up * 1e3, "upload_gb_per_s": host.nbytes / t_up / 1e9,
        # (the first upload makes the staging buffers and the scan's device memory; the second is the rate of the link)
        "ms_upload_again": t_up2 * 1e3, "upload_again_gb_per_s": host.nbytes / t_up2 / 1e9}
    del host, res, res_q
    # the passes over the binned scan: each reads x, y of every window member (16 B) once per sweep
    del xyz, ws
    torch.cuda.empty_cache()
    tot = int(off[-1].item())
    longest = int((off[1:] - off[:-1]).max().item())
    kasa, fit = (torch.empty(S, 3, dtype=torch.float64, device=dev) for _ in range(2))
    iters = torch.empty(S, 4, dtype=torch.int32, device=dev)
    st = torch.empty(S, dtype=torch.int32, device=dev)
    ms_fit = timed(lambda: F.check(L.pcfit_fit_batch_dev(ptr(xo), ptr(yo), ptr(off), S, longest, 1.0, 100, 1e-2, 1e-10,
                                                         1e-12, ptr(kasa), ptr(fit), ptr(iters), ptr(st), stream)), 10)
    sweeps = float((iters[:, 2] + iters[:, 3]).double().mean().item())
    sel = torch.empty(S, 720, dtype=torch.int64, device=dev)
    ms_sel = timed(lambda: F.check(L.pcfit_angle_max_batch_dev(ptr(xo), ptr(yo), ptr(off), S, ptr(fit), 720, ptr(sel),
                                                               stream)), 10)
    H = 256
    edges = torch.linspace(-0.5, 0.5, H + 1, dtype=torch.float64, device=dev)
    hist = torch.empty(S, H, dtype=torch.int64, device=dev)
    rmm = torch.empty(S, 2, dtype=torch.float64, device=dev)
    amm = torch.empty(S, 2, dtype=torch.int64, device=dev)
    ms_hist = timed(lambda: F.check(L.pcfit_residuals_batch_dev(ptr(xo), ptr(yo), ptr(off), S, ptr(fit), None, None, None,
                                                                H, ptr(edges), ptr(hist), ptr(rmm), ptr(amm), stream)), 10)
    # both from one read of every window (pcfit_boundary_residuals_batch_dev), and that they are the same numbers
    sel1, hist1, amm1 = torch.empty_like(sel), torch.empty_like(hist), torch.empty_like(amm)
    ms_both = timed(lambda: F.check(L.pcfit_boundary_residuals_batch_dev(ptr(xo), ptr(yo), ptr(off), S, ptr(fit), 720,
                                                                         ptr(sel1), H, ptr(edges), ptr(hist1), ptr(rmm),
                                                                         ptr(amm1), stream)), 10)
    same = bool(torch.equal(sel, sel1) and torch.equal(hist, hist1) and torch.equal(amm, amm1))
    gbs = lambda ms_, passes: 16.0 * tot * passes / (ms_ * 1e-3) / 1e9
    out["boundary_and_histograms_100M_points_2000_windows"] = {
        "ms_binning": ms, "ms_kasa_lm_fit": ms_fit, "mean_lm_sweeps": sweeps,
        "ms_angular_bin_boundary_720": ms_sel, "ms_residual_histogram_256": ms_hist,
        "ms_boundary_and_histogram_one_sweep": ms_both, "one_sweep_equals_the_two_calls": same,
        "hbm_gbs_one_sweep": gbs(ms_both, 1), "hbm_frac_one_sweep": gbs(ms_both, 1) / PEAK_HBM_GBS,
        "ms_total": ms + ms_fit + ms_both, "points_per_s": n / ((ms + ms_fit + ms_both) * 1e-3),
        "longest_window": longest, "all_fits_ok": bool((st == 0).all().item()),
        "hbm_gbs_boundary": gbs(ms_sel, 1), "hbm_frac_boundary": gbs(ms_sel, 1) / PEAK_HBM_GBS,
        "hbm_gbs_histogram": gbs(ms_hist, 1), "hbm_frac_histogram": gbs(ms_hist, 1) / PEAK_HBM_GBS,
        "hbm_gbs_fit_per_sweep": gbs(ms_fit, sweeps + 1), "hbm_frac_fit_per_sweep": gbs(ms_fit, sweeps + 1) / PEAK_HBM_GBS}
    return out


def launch_ranks(n, argv):
    """`bench.py --gpus N` started plainly: run the N ranks as a child `torch.distributed.run` (this process has
    not touched the GPU yet and never will) and pass rank 0's line through."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--slices", type=int, default=10_000)
    ap.add_argument("--points", type=int, default=8192)
    ap.add_argument("--bins", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the timings of the other configurations")
    ap.add_argument("--no-side", action="store_true",
                    help="skip the timings of the subsystems alone (a profile then holds the timed launches only)")
    ap.add_argument("--backend", default="nccl", help="nccl (RCCL) on GPUs; gloo only rehearses the launch")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # (more ranks than GPUs only happens in a rehearsal of the launch on a smaller box, with --backend gloo)
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # The fit kernels run on a stream of their own at normal priority and the process group's stream is asked for
    # at high priority: HIP gives streams of different priorities different hardware queues. Two streams that land
    # on one hardware queue run their work in submission order, and the gather of step i then sits in front of the
    # fit kernel of step i+1 instead of beside it (seen in a kernel trace: 45 us between fit kernels, 20 us apart).
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    dist = None
    # PCFIT_BENCH_FORCE_DIST=1 runs the collective path in a one-rank group (rehearsal on a one-GPU box)
    use_dist = world > 1 or os.environ.get("PCFIT_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
        if a.backend == "nccl":
            opts = dist.ProcessGroupNCCL.Options()
            opts.is_high_priority_stream = True
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world, pg_options=opts)
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)
    L = F.lib()

    S, n, H = a.slices, a.points, a.bins
    sigma = 0.05
    lo, hi = -10 * sigma, 10 * sigma
    xh, yh, offh = make_slices(S, n, seed=1 + rank, sigma=sigma, jitter=0.02)
    x = torch.from_numpy(xh).to(dev)
    y = torch.from_numpy(yh).to(dev)
    off = torch.from_numpy(offh).to(dev)
    edges = torch.from_numpy(np.linspace(lo, hi, H + 1)).to(dev)
    kasa = torch.empty(S, 3, dtype=torch.float64, device=dev)
    hist = torch.empty(S, H, dtype=torch.int64, device=dev)
    # What travels between the ranks, one packed record per step: the S fitted circles (f64) followed by the
    # S x H histogram counts. A count never exceeds the slice length, so 16 bits hold it for slices of up to
    # 32767 points (half the bytes of int32 on the links). The fit kernel writes its circles straight into the
    # record, so one collective per step carries everything and only the counts need a narrowing copy.
    wire = torch.int16 if n <= 32767 else torch.int32
    fit_bytes = S * 3 * 8
    rec_bytes = fit_bytes + S * H * wire.itemsize
    record = [torch.empty(rec_bytes, dtype=torch.uint8, device=dev) for _ in range(RING)]
    fit = [r[:fit_bytes].view(torch.float64).view(S, 3) for r in record]
    hist_wire = [r[fit_bytes:].view(wire).view(S, H) for r in record]
    iters = torch.empty(S, 4, dtype=torch.int32, device=dev)
    status = torch.empty(S, dtype=torch.int32, device=dev)
    rmm = torch.empty(S, 2, dtype=torch.float64, device=dev)
    amm = torch.empty(S, 2, dtype=torch.int64, device=dev)
    on_host = use_dist and a.backend != "nccl"          # rehearsal: the exchange goes through host memory
    if use_dist:
        gathered = [torch.empty(world * rec_bytes, dtype=torch.uint8, device="cpu" if on_host else dev)
                    for _ in range(RING)]
    pending = [None] * RING
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ev_k = []

    def step(i, timed):
        b = i % RING
        # Before a record buffer is written again its previous gather must have finished. Collectives complete in
        # issue order, so one wait every RING/2 steps on the gather issued RING/2 steps ago covers the next RING/2
        # buffers; a wait per step would put a cross-queue barrier in front of every fit kernel.
        if b % (RING // 2) == 0:
            w = pending[(b + RING // 2) % RING]
            if w is not None:
                w.wait()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        F.check(L.pcfit_fit_residuals_batch_dev(ptr(x), ptr(y), ptr(off), S, n, 1.0, 100, 1e-2, 1e-10, 1e-12,
                                                ptr(kasa), ptr(fit[b]), ptr(iters), ptr(status), H, ptr(edges),
                                                ptr(hist), ptr(rmm), ptr(amm), stream))
        e1.record()
        if timed:
            ev_k.append((e0, e1))
        if use_dist:
            hist_wire[b].copy_(hist)
            # (bytes on the wire: RCCL has no 16-bit integer type)
            if on_host:
                dist.all_gather_into_tensor(gathered[b], record[b].cpu())
            else:
                pending[b] = dist.all_gather_into_tensor(gathered[b], record[b], async_op=True)

    def drain():
        for b in range(RING):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    # An idle GPU takes some 20 ms of load to bring its clock up (launch durations from idle on this part: 0.55 ms
    # at the 10th launch, 0.50 at the 30th, 0.48 from the 40th on), so the kernel runs SETTLE times before the warmup
    # steps: what is timed is the rate of a GPU at work, and the line names what the first launches took.
    settle_ev = []
    if use_dist:
        dist.barrier()          # (the ranks get here seconds apart: they settle, warm up and reach the timed region together)
    for i in range(SETTLE):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        F.check(L.pcfit_fit_residuals_batch_dev(ptr(x), ptr(y), ptr(off), S, n, 1.0, 100, 1e-2, 1e-10, 1e-12,
                                                ptr(kasa), ptr(fit[0]), ptr(iters), ptr(status), H, ptr(edges),
                                                ptr(hist), ptr(rmm), ptr(amm), stream))
        e1.record()
        settle_ev.append((e0, e1))
    for i in range(a.warmup):
        step(i, False)
    drain()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i, True)
    t_host = time.perf_counter() - t0
    drain()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if os.environ.get("PCFIT_BENCH_DEBUG") == "1":
        print(f"[bench] host enqueue {t_host / a.steps * 1e3:.3f} ms/step of {dt / a.steps * 1e3:.3f}", file=sys.stderr)
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if on_host else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # sweeps of the timed launches, read before anything else writes `iters`
    it = iters.cpu().numpy().astype(np.float64)

    # the subsystems on their own, after the timed region: algebraic fit, fit without residuals, residuals
    def timed(fn, reps=20):
        settle(fn)
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    side = []
    if rank == 0 and not a.no_side:
        iters_side = torch.empty_like(iters)
        fit_side = torch.empty(S, 3, dtype=torch.float64, device=dev)
        fit_side.copy_(fit[(a.warmup + a.steps - 1) % RING] if a.steps + a.warmup > 0 else fit[0])
        for name, fn in (
            ("kasa_kernel (algebraic fit)", lambda: F.check(L.pcfit_kasa_batch_dev(
                ptr(x), ptr(y), ptr(off), S, n, ptr(kasa), ptr(status), stream))),
            ("fit_kernel (Kasa + LM)", lambda: F.check(L.pcfit_fit_batch_dev(
                ptr(x), ptr(y), ptr(off), S, n, 1.0, 100, 1e-2, 1e-10, 1e-12, ptr(kasa), ptr(fit_side), ptr(iters_side),
                ptr(status), stream))),
            ("fit_residual_kernel without the LM loop (max_iter = 0: Kasa + residual pass, one read of the slice)",
             lambda: F.check(L.pcfit_fit_residuals_batch_dev(
                 ptr(x), ptr(y), ptr(off), S, n, 1.0, 0, 1e-2, 1e-10, 1e-12, ptr(kasa), ptr(fit_side), ptr(iters_side),
                 ptr(status), H, ptr(edges), ptr(hist), ptr(rmm), ptr(amm), stream))),
            ("residuals_kernel (histogram + extremes)", lambda: F.check(L.pcfit_residuals_batch_dev(
                ptr(x), ptr(y), ptr(off), S, ptr(fit_side), None, None, None, H, ptr(edges), ptr(hist), ptr(rmm),
                ptr(amm), stream))),
        ):
            ms = timed(fn)
            side.append({"kernel": name, "ms": ms, "hbm_gbs": S * n * 16 / (ms * 1e-3) / 1e9,
                         "hbm_frac": S * n * 16 / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS})

    ms_step = dt / a.steps * 1e3
    settle_ms = [s.elapsed_time(e) for s, e in settle_ev]
    per_launch = [s.elapsed_time(e) for s, e in ev_k]
    k_ms = float(np.mean(per_launch))
    if os.environ.get("PCFIT_BENCH_DEBUG") == "1" and rank == 0:
        print("[bench] launch durations (ms), every 10th: " + " ".join(f"{v:.3f}" for v in per_launch[::10]), file=sys.stderr)
    sweeps = it[:, 2] + it[:, 3]
    flop_launch = float(n * (FLOP_SWEEP_FIRST * it[:, 2].sum() + FLOP_SWEEP * it[:, 3].sum() + (FLOP_KASA + FLOP_RESID) * S))
    alg_bytes = float(S * n * 16)

    if rank == 0:
        traffic = {}
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath))
        # counters of an earlier profiler run (tools/profile_bench.sh + tools/summarize_profile.py): only if the
        # kernel's sources are still the ones they were measured on
        fresh = bool(traffic) and traffic.get("kernel_sources_sha1") == kernel_sources_sha1()
        tf = flop_launch / (k_ms * 1e-3) / 1e12
        out = {
            "metric": "points fitted/sec (circle fit+residuals)",
            "value": world * S * n / (dt / a.steps),
            "unit": "points/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": ms_step,
            # launches before the warmup steps (untimed): the clock of an idle GPU comes up over the first ~40 of them
            "clock_settling": {"launches_before_warmup": SETTLE,
                               "ms_per_launch_2nd_to_21st_from_idle": float(np.mean(settle_ms[1:21])) if len(settle_ms) > 21 else None,
                               "ms_per_launch_last_10_of_them": float(np.mean(settle_ms[-10:])) if settle_ms else None},
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{S} slices x {n} points per GPU, Kasa+LM/Huber fit then residual "
                                   f"histogram ({H} bins) + Rmax/Rmin, pre-binned CSR, noisy cylinder sigma=0.05",
                       "slices_per_gpu": S, "points_per_slice": n, "hist_bins": H,
                       "parallelism": f"slice-sharded x{world}" + ((", RCCL all-gather of fits+histograms" if a.backend == "nccl" else
                                                                         f", {a.backend} all-gather through host memory (launch rehearsal)")
                                                                        if world > 1 else "")},
            # The metric asks for the fraction of the HBM roofline; the pass itself is bound by FP64 vector
            # issue (see roofline_valu), so this fraction cannot approach 1: two LM sweeps, the Kasa power
            # sums and the residual decisions are ~140 vector instructions per point of 16 bytes.
            "roofline": {
                "kernel": "fit_residual_kernel<16,16,256> (Kasa + LM + residuals, one read of the slice)",
                "bound": "hbm",
                "achieved": alg_bytes / (k_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": alg_bytes / (k_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                "traffic": traffic.get("fit_residual_kernel_bytes") if fresh else None,
                "traffic_source": (traffic.get("source") if fresh else
                                   "none: profiles/traffic.json was measured on other kernel sources" if traffic else None),
                "avg_launch_ms": k_ms,
                "algorithmic_bytes": alg_bytes,
            },
            "roofline_valu": {
                "kernel": "fit_residual_kernel<16,16,256>", "bound": "valu_fp64",
                "achieved": tf, "peak": PEAK_FP64_VALU_TFLOPS, "unit": "TFLOP/s",
                "frac": tf / PEAK_FP64_VALU_TFLOPS,
                "flop_per_point": flop_launch / (S * n),
                "mean_sweeps": float(sweeps.mean()),
                "avg_launch_ms": k_ms,
                # The peak is 256 CUs x 64 lanes x 2 flop at 2.4 GHz; under this kernel the counters show a clock near
                # 2.0 GHz (SQ_BUSY_CYCLES / 32 engines / launch time), so a kernel that filled every issue slot with
                # an FMA would read 0.82 here.  Below: the share of the SIMDs' cycles in which this kernel issued a
                # vector instruction and the instructions it issued per point, from the last profiler run on these
                # sources (null when the sources have changed since).
                "valu_issue_frac_last_pmc": traffic.get("fit_residual_kernel_valu_issue_frac") if fresh else None,
                "valu_insts_per_point_last_pmc": traffic.get("fit_residual_kernel_valu_insts_per_point") if fresh else None,
            },
            "kernels": side,
        }
        # the pass the 50 %-of-HBM target is stated for: algebraic fit + residual statistics, one read of the slice
        if side:
            kr = side[2]
            out["roofline_kasa_residual_pass"] = {
                "kernel": "fit_residual_kernel<22,10,256,false> (max_iter = 0: no LM loop, three workgroups to a CU)",
                "bound": "hbm", "achieved": kr["hbm_gbs"],
                "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": kr["hbm_frac"], "avg_launch_ms": kr["ms"],
                "algorithmic_bytes": alg_bytes}
        if not a.no_extra:
            out["extra"] = extra_configs(L, dev, stream)
        if world == 1 and not a.no_cpu_baseline:
            host_cpus = os.cpu_count() or 1
            # (on a shared box os.cpu_count() is the whole machine; a job's share is what OMP_NUM_THREADS says)
            threads = max(1, min(host_cpus, int(os.environ.get("OMP_NUM_THREADS", "16") or 16), 64))
            rate, k, secs = cpu_baseline(xh, yh, offh, H, lo, hi)
            out["cpu_baseline"] = {"value": rate, "unit": "points/s", "cores": 1, "kind": "port",
                                   "host_cpu_count": host_cpus,
                                   "sample": f"numpy restatement of the same step on the first {k} slices "
                                             f"({k * n} points, {secs:.1f} s, one Python thread, the reference's "
                                             f"serial loop analyzer.py:148)"}
            rate_p, kp, secs_p = cpu_baseline_pool(xh, yh, offh, H, lo, hi, threads, max(4 * threads, k // 2))
            out["cpu_baseline_pool"] = {"value": rate_p, "unit": "points/s", "cores": threads, "kind": "port",
                                        "host_cpu_count": host_cpus,
                                        "sample": f"the same through ThreadPoolExecutor({threads}) as analyzer.py:203-207 "
                                                  f"on {kp} slices, {secs_p:.1f} s"}
            rate_c, kc, secs_c = cpu_baseline_c(xh, yh, offh, H, lo, hi)
            out["cpu_baseline_c"] = {"value": rate_c, "unit": "points/s", "cores": 1, "kind": "port",
                                     "sample": f"plain C restatement (literal 100 iterations, no interpreter) on {kc} "
                                               f"slices, {secs_c:.1f} s"}
            rate_ct, kct, secs_ct = cpu_baseline_c(xh, yh, offh, H, lo, hi, n_slices=600 * threads, threads=threads)
            out["cpu_baseline_c_threads"] = {"value": rate_ct, "unit": "points/s", "cores": threads, "kind": "port",
                                             "sample": f"the C restatement on {threads} host threads, {kct} slices, "
                                                       f"{secs_ct:.1f} s"}
            out["gpu_over_cpu"] = out["value"] / rate
            out["gpu_over_best_cpu"] = out["value"] / max(rate, rate_p, rate_c, rate_ct)
        print(json.dumps(out))
    if use_dist:
        # rank 0 spends some seconds in the extras above: nobody takes the group down before it is back
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
