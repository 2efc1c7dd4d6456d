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

# How the flop figure of the kernel is counted: the arithmetic of the short form of each pass, per
# point, an FMA as 2, everything else (add, mul, compare, min, the hardware reciprocal square root) as 1.
#   Kasa: sums (2) + centred moments (2 sub, 2 mul, 2 add, 5 fma)                               = 18
#   LM sweep: distance (2 sub, mul, fma), rsqrt + 2 Newton steps (10), ri, 1/ri (5), residual and
#             its test (2), Jacobian (2 mul), 7 fma + 3 add into H, g, cost, curvature, min (1)   = 42
#   residuals: exact sum of squares (13), rsqrt + Newton (10), correctly rounded root (9), dr,
#              extremes (5), bin guess and repair (8)                                             = 45
# The reference's literal loop evaluates residuals and Jacobian separately in each of its 100
# iterations (~1e4 flop per point); the kernel reaches the same accept/reject sequence in 3 sweeps.
FLOP_KASA = 18.0
FLOP_SWEEP = 42.0
FLOP_RESID = 45.0
PEAK_FP64_VALU_TFLOPS = 78.6
RING = 8                      # record buffers in flight between the fit kernel and the all-gather
PEAK_HBM_GBS = 8000.0


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def cpu_baseline(x, y, off, bins, lo, hi, budget_s=14.0, max_slices=2000):
    """numpy restatement of the same step (Kasa -> LM -> residual histogram), slice after slice,
    on the host: the stand-in for the reference's CPU path, which cannot travel."""
    from oracle import ref_np as R
    t0 = time.perf_counter()
    done = pts = 0
    while done < max_slices and time.perf_counter() - t0 < budget_s:
        sl = slice(off[done], off[done + 1])
        xs, ys = x[sl], y[sl]
        k = R.kasa_fit(xs, ys)
        f = R.lm_huber(xs, ys, *k)
        theta, r = R.polar(np.stack([xs, ys], 1), f[0], f[1])
        R.residual_histogram(r - f[2], bins, lo, hi)
        int(np.argmax(r)), int(np.argmin(r))
        done += 1
        pts += xs.size
    dt = time.perf_counter() - t0
    return pts / dt, done, dt


def cpu_baseline_c(x, y, off, bins, lo, hi, n_slices=600):
    from oracle import native as N
    n_slices = min(n_slices, len(off) - 1)
    end = off[n_slices]
    t0 = time.perf_counter()
    N.fit_residual_batch(x[:end], y[:end], off[:n_slices + 1], bins, lo, hi)
    dt = time.perf_counter() - t0
    return end / dt, n_slices, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--slices", type=int, default=10_000)
    ap.add_argument("--points", type=int, default=8192)
    ap.add_argument("--bins", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
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
        opts = dist.ProcessGroupNCCL.Options()
        opts.is_high_priority_stream = True
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world, pg_options=opts)
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
    if use_dist:
        gathered = [torch.empty(world * rec_bytes, dtype=torch.uint8, device=dev) for _ in range(RING)]
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
            pending[b] = dist.all_gather_into_tensor(gathered[b], record[b], async_op=True)

    def drain():
        for b in range(RING):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

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
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # the subsystems on their own, after the timed region: algebraic fit, fit without residuals, residuals
    def timed(fn, reps=5):
        fn()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    side = []
    if rank == 0:
        for name, fn in (
            ("kasa_kernel (algebraic fit)", lambda: F.check(L.pcfit_kasa_batch_dev(
                ptr(x), ptr(y), ptr(off), S, n, ptr(kasa), ptr(status), stream))),
            ("fit_kernel (Kasa + LM)", lambda: F.check(L.pcfit_fit_batch_dev(
                ptr(x), ptr(y), ptr(off), S, n, 1.0, 100, 1e-2, 1e-10, 1e-12, ptr(kasa), ptr(fit[0]), ptr(iters),
                ptr(status), stream))),
            ("fit_residual_kernel with max_iter = 0 (Kasa + residual pass, one read of the slice)",
             lambda: F.check(L.pcfit_fit_residuals_batch_dev(
                 ptr(x), ptr(y), ptr(off), S, n, 1.0, 0, 1e-2, 1e-10, 1e-12, ptr(kasa), ptr(fit[0]), ptr(iters),
                 ptr(status), H, ptr(edges), ptr(hist), ptr(rmm), ptr(amm), stream))),
            ("residuals_kernel (histogram + extremes)", lambda: F.check(L.pcfit_residuals_batch_dev(
                ptr(x), ptr(y), ptr(off), S, ptr(fit[0]), None, None, None, H, ptr(edges), ptr(hist), ptr(rmm),
                ptr(amm), stream))),
        ):
            ms = timed(fn)
            side.append({"kernel": name, "ms": ms, "hbm_gbs": S * n * 16 / (ms * 1e-3) / 1e9,
                         "hbm_frac": S * n * 16 / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS})

    ms_step = dt / a.steps * 1e3
    k_ms = float(np.mean([s.elapsed_time(e) for s, e in ev_k]))
    it = iters.cpu().numpy().astype(np.float64)
    sweeps = it[:, 2] + it[:, 3]
    flop_launch = float(n * (FLOP_SWEEP * sweeps.sum() + (FLOP_KASA + FLOP_RESID) * S))
    alg_bytes = float(S * n * 16)

    if rank == 0:
        traffic = {}
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath))
        tf = flop_launch / (k_ms * 1e-3) / 1e12
        out = {
            "metric": "points fitted/sec (circle fit+residuals)",
            "value": world * S * n / (dt / a.steps),
            "unit": "points/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{S} slices x {n} points per GPU, Kasa+LM/Huber fit then residual "
                                   f"histogram ({H} bins) + Rmax/Rmin, pre-binned CSR, noisy cylinder sigma=0.05",
                       "slices_per_gpu": S, "points_per_slice": n, "hist_bins": H,
                       "parallelism": f"slice-sharded x{world}" + (", RCCL all-gather of fits+histograms" if world > 1 else "")},
            # The metric asks for the fraction of the HBM roofline; the pass itself is bound by FP64 vector
            # issue (see roofline_valu), so this fraction cannot approach 1: three LM sweeps, the Kasa
            # moments and a correctly rounded hypot per point are ~220 vector instructions per 16 bytes.
            "roofline": {
                "kernel": "fit_residual_kernel<16,16,256> (Kasa + LM + residuals, one read of the slice)",
                "bound": "hbm",
                "achieved": alg_bytes / (k_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": alg_bytes / (k_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                "traffic": traffic.get("fit_residual_kernel_bytes"),
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
            },
            "kernels": side,
        }
        if world == 1 and not a.no_cpu_baseline:
            rate, k, secs = cpu_baseline(xh, yh, offh, H, lo, hi)
            out["cpu_baseline"] = {"value": rate, "unit": "points/s", "cores": 1, "kind": "port",
                                   "sample": f"numpy restatement of the same step on the first {k} slices "
                                             f"({k * n} points, {secs:.1f} s, one Python thread)"}
            rate_c, kc, secs_c = cpu_baseline_c(xh, yh, offh, H, lo, hi)
            out["cpu_baseline_c"] = {"value": rate_c, "unit": "points/s", "cores": 1, "kind": "port",
                                     "sample": f"plain C restatement (literal 100 iterations) on {kc} slices, {secs_c:.1f} s"}
            out["gpu_over_cpu"] = out["value"] / rate
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
