"""Slice sharding over the GPUs of one node.

Slices are independent (each window reads only its own points), so the path
shards with no exchange of points: every rank takes a contiguous range of
windows, balanced by point count, and only per-slice results travel - fitted
parameters and residual histograms, all-gathered with torch.distributed
(backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in CPU tests).
"""
from __future__ import annotations

import numpy as np

from .analyzer import _FIT


def split_by_points(counts, world: int) -> np.ndarray:
    """Bounds (world + 1 entries) of contiguous slice ranges whose point totals
    are as even as the slice sizes allow: rank r owns slices bounds[r]:bounds[r+1].
    Cuts sit where the running point total crosses r/world of the grand total."""
    counts = np.asarray(counts, dtype=np.int64)
    S = counts.size
    bounds = np.zeros(world + 1, dtype=np.int64)
    bounds[world] = S
    if S == 0:
        return bounds
    csum = np.cumsum(counts)
    total = int(csum[-1])
    for r in range(1, world):
        if total == 0:
            cut = (S * r) // world
        else:
            # first slice whose inclusive running total exceeds the target ends the previous rank
            target = total * r / world
            cut = int(np.searchsorted(csum, target, side="left"))
            # take the slice across the boundary if most of it lies before the target
            if cut < S and (csum[cut] - target) < (target - (csum[cut] - counts[cut])):
                cut += 1
        bounds[r] = min(max(cut, bounds[r - 1]), S)
    return bounds


def shard_range(bounds, rank: int):
    return int(bounds[rank]), int(bounds[rank + 1])


def allgather_slice_arrays(local, bounds, group=None):
    """All-gather per-slice arrays.  ``local``: list of torch tensors whose first
    dimension is this rank's slice count.  Returns the same list for ALL slices, in
    global slice order, on every rank.  Shards are padded to the largest one so
    that a single fixed-size all_gather per array does the job."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    sizes = np.diff(np.asarray(bounds, dtype=np.int64))
    assert sizes.size == world
    pad = int(sizes.max()) if world else 0
    out = []
    for t in local:
        k = t.shape[0]
        assert k == sizes[dist.get_rank(group)], "local array does not match this rank's shard"
        buf = torch.zeros((pad,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        buf[:k] = t
        gathered = torch.empty((world * pad,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(gathered, buf, group=group)
        parts = [gathered[r * pad:r * pad + int(sizes[r])] for r in range(world)]
        out.append(torch.cat(parts, dim=0))
    return out


REC = 16        # doubles per window in the exchanged record (13 used; a batch of scans adds the window's centre
                # and the number of its scan at 13 and 14)


def pack_window_records(h):
    """Per-window results of a hybrid batch as one (S, REC) float64 array - population, status, candidate
    count, arg-max / arg-min candidate, fitted circle, the five ovality scalars - and the candidates as one
    (2, n_cand) array.  Integers are exact in float64 (all far below 2^53)."""
    S = len(h["status"])
    rec = np.zeros((S, REC))
    rec[:, 0] = np.diff(h["offsets"])
    rec[:, 1] = h["status"]
    rec[:, 2] = np.diff(h["cand_offsets"])
    rec[:, 3] = h["argmax"]
    rec[:, 4] = h["argmin"]
    rec[:, 5:8] = h["fit"]
    rec[:, 8:13] = h["ov"]
    return rec, np.stack([h["cand_x"], h["cand_y"]])


def unpack_window_records(rec, cand):
    """Inverse of pack_window_records for any number of windows: the dict CylinderAnalyzer._window_results reads."""
    S = rec.shape[0]
    n0 = rec[:, 0].astype(np.int64)
    nc = rec[:, 2].astype(np.int64)
    offsets = np.zeros(S + 1, np.int64)
    np.cumsum(n0, out=offsets[1:])
    co = np.zeros(S + 1, np.int64)
    np.cumsum(nc, out=co[1:])
    return dict(offsets=offsets, status=rec[:, 1].astype(np.int32), cand_offsets=co,
                cand_x=np.ascontiguousarray(cand[0]), cand_y=np.ascontiguousarray(cand[1]),
                argmax=rec[:, 3].astype(np.int64), argmin=rec[:, 4].astype(np.int64),
                fit=np.ascontiguousarray(rec[:, 5:8]), ov=np.ascontiguousarray(rec[:, 8:13]))


def _collective_device(group):
    """Where tensors of a collective must live: the calling thread's GPU under RCCL, host memory otherwise."""
    import torch
    import torch.distributed as dist
    if dist.get_backend(group) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def allgather_window_records(rec, cand, bounds, group=None):
    """Two fixed-size all-gathers (all_gather_into_tensor: RCCL on GPUs): the window records of every rank,
    padded to the largest shard, then - their sizes known from the records - the candidates, padded to the
    largest total.  Returns (records of all windows in window order, candidates (2, total))."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = np.diff(np.asarray(bounds, dtype=np.int64))
    dev = _collective_device(group)
    pad = max(int(sizes.max()), 1)
    mine = torch.zeros((pad, REC), dtype=torch.float64)
    mine[:rec.shape[0]] = torch.from_numpy(rec)
    mine = mine.to(dev)
    everyone = torch.empty((world * pad, REC), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(everyone, mine, group=group)
    everyone = everyone.cpu().numpy().reshape(world, pad, REC)
    recs = np.concatenate([everyone[r, :int(sizes[r])] for r in range(world)], axis=0)
    totals = [int(everyone[r, :int(sizes[r]), 2].sum()) for r in range(world)]
    cpad = max(max(totals), 1)
    mine_c = torch.zeros((2, cpad), dtype=torch.float64)
    mine_c[:, :cand.shape[1]] = torch.from_numpy(np.ascontiguousarray(cand))
    mine_c = mine_c.to(dev)
    every_c = torch.empty((world * 2, cpad), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(every_c, mine_c, group=group)
    every_c = every_c.cpu().numpy().reshape(world, 2, cpad)
    cands = np.concatenate([every_c[r, :, :totals[r]] for r in range(world)], axis=1)
    return recs, cands


def analyze_sharded(analyzer, points, group=None, progress_callback=None):
    """CylinderAnalyzer.analyze_with_progress with the windows sharded over the
    ranks of ``group``: every rank takes a contiguous range of windows balanced
    by point count, uploads ONLY the rows of the scan whose z lies in that range
    (closed edges: a point on the edge shared with the neighbouring rank goes to
    both), bins and fits them on its GPU, and the per-window records (fit,
    metrics, candidates) are exchanged with two fixed-size all-gathers, so that
    every rank returns the full, z-ordered list - the same dicts, bit for bit, as
    a single rank computes.  ``points`` is the whole scan on every rank, as an
    (N, 3) array, or a DeviceScan already on the rank's GPU.  The work runs on
    the device that is current in the calling thread (one process per GPU: select
    it with torch.cuda.set_device(local_rank) or pcfit_set_device before the
    call)."""
    import torch.distributed as dist

    from . import _ffi
    from ._ffi import DeviceScan

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dz = analyzer.config.z_step / 2.0
    ranged = analyzer.z_min is not None and analyzer.z_max is not None
    resident = isinstance(points, DeviceScan)
    if resident:
        if ranged:
            raise ValueError("a DeviceScan cannot be combined with Config.z_min / z_max: pass the array")
        z_centers = analyzer.calculate_z_centers(points)
        counts = (points.window_counts(z_centers - dz, z_centers + dz) if len(z_centers) and len(points)
                  else np.zeros(len(z_centers), np.int64))
    else:
        points = np.ascontiguousarray(points, dtype=np.float64)
        if points.ndim != 2 or points.shape[1] != 3:
            raise ValueError("Points must be Nx3 array")
        if ranged:
            z_lo_all, z_hi_all = analyzer.z_min, analyzer.z_max
        elif len(points):
            # (calculate_z_centers would make two strided numpy passes over the z column)
            z_lo_all, z_hi_all = (np.float64(v) for v in _ffi.host_z_range(points))
        else:
            z_lo_all = z_hi_all = np.float64("nan")
        z_centers = np.arange(z_lo_all + dz, z_hi_all - dz + 1e-9, analyzer.config.z_step)
        counts = (_ffi.host_window_counts(points, z_centers - dz, z_centers + dz) if len(z_centers) and len(points)
                  else np.zeros(len(z_centers), np.int64))
        if ranged and counts.sum() == 0:
            # the reference filters by the configured range first and refuses an empty result (analyzer.py:190-196)
            z = points[:, 2]
            if not np.any((z >= analyzer.z_min) & (z <= analyzer.z_max)):
                raise ValueError(f"No points in specified Z range [{analyzer.z_min}, {analyzer.z_max}]")
    S = len(z_centers)
    bounds = split_by_points(counts, world)
    s0, s1 = shard_range(bounds, rank)
    empty = dict(offsets=np.zeros(1, np.int64), status=np.zeros(0, np.int32), cand_offsets=np.zeros(1, np.int64),
                 cand_x=np.zeros(0), cand_y=np.zeros(0), argmax=np.zeros(0, np.int64), argmin=np.zeros(0, np.int64),
                 fit=np.zeros((0, 3)), ov=np.zeros((0, 5)))
    h = empty
    if s1 > s0:
        lo, hi = z_centers[s0:s1] - dz, z_centers[s0:s1] + dz
        args = (_FIT["inlier_tol"], _FIT["max_trials"], _FIT["min_inliers_frac"], None, _FIT["huber_delta"],
                _FIT["use_hull"])
        if resident:
            h = _ffi.scan_hybrid_batch(points, lo, hi, *args, order=analyzer.config.hull_order)
        else:
            # (with a configured range the reference drops what lies outside it first, analyzer.py:191-196: the last
            # window's upper edge can lie a rounding above z_max, and a row in between is nobody's)
            z_range = (max(lo[0], analyzer.z_min), min(hi[-1], analyzer.z_max)) if ranged else (lo[0], hi[-1])
            with DeviceScan(points, z_range=z_range) as part:
                if len(part):
                    h = _ffi.scan_hybrid_batch(part, lo, hi, *args, order=analyzer.config.hull_order)
                else:
                    h = dict(empty, offsets=np.zeros(s1 - s0 + 1, np.int64), status=np.ones(s1 - s0, np.int32),
                             cand_offsets=np.zeros(s1 - s0 + 1, np.int64), argmax=np.zeros(s1 - s0, np.int64),
                             argmin=np.zeros(s1 - s0, np.int64), fit=np.full((s1 - s0, 3), np.nan),
                             ov=np.full((s1 - s0, 5), np.nan))
    rec, cand = pack_window_records(h)
    recs, cands = allgather_window_records(rec, cand, bounds, group)
    assert recs.shape[0] == S
    results = [res for res, _, warn in analyzer._window_results(z_centers, unpack_window_records(recs, cands), False)
               if warn is None]
    results.sort(key=lambda r: r["z_center"])
    if progress_callback:
        progress_callback(100)
    return results


def allgather_scan_records(mine, group=None):
    """The per-window results of whole scans, exchanged between the ranks.  ``mine``: {scan number: (z_centers,
    h)} of the scans this rank analysed (h as _ffi.scan_hybrid_batch returns it; scans without windows left out).
    Returns the same for the scans of ALL ranks.  One gather of every rank's window count, then the two
    fixed-size gathers of allgather_window_records; the record of a window carries its centre and the number of
    its scan, so the receiver needs nothing else to take the stream apart."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    recs, cands = [], []
    for k in sorted(mine):
        z_centers, h = mine[k]
        rec, cand = pack_window_records(h)
        rec[:, 13] = z_centers
        rec[:, 14] = k
        recs.append(rec)
        cands.append(cand)
    rec = np.concatenate(recs, axis=0) if recs else np.zeros((0, REC))
    cand = np.concatenate(cands, axis=1) if cands else np.zeros((2, 0))
    dev = _collective_device(group)
    count = torch.tensor([rec.shape[0]], dtype=torch.int64, device=dev)
    counts = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, count, group=group)
    bounds = np.concatenate([[0], np.cumsum(counts.cpu().numpy())]).astype(np.int64)
    all_rec, all_cand = allgather_window_records(rec, cand, bounds, group)
    # the windows of one scan stand together (its owner sent them in window order): cut where the number changes
    scan_of = all_rec[:, 14].astype(np.int64)
    c_off = np.concatenate([[0], np.cumsum(all_rec[:, 2].astype(np.int64))])
    starts = np.flatnonzero(np.diff(scan_of, prepend=-1)) if scan_of.size else np.zeros(0, np.int64)
    ends = np.append(starts[1:], scan_of.size)
    out = {}
    for a, b in zip(starts, ends):
        out[int(scan_of[a])] = (all_rec[a:b, 13].copy(),
                                unpack_window_records(all_rec[a:b], all_cand[:, c_off[a]:c_off[b]]))
    return out


def scans_of_rank(n_scans: int, rank: int, world: int):
    """Whole scans go to the ranks round-robin: rank r analyses scans r, r + world, ..."""
    return list(range(rank, n_scans, world))


def analyze_scans_sharded(analyzer, scans, group=None, progress_callback=None, gather=True):
    """A batch of independent scans over the ranks of ``group``: whole scans round-robin to the ranks (no cuts
    through a scan, no exchange of points), each rank running its scans one after the other with the upload of
    the next one (a second host thread making the DeviceScan) overlapped with the analysis of the current one.
    ``scans``: a sequence whose items are (N, 3) arrays or callables returning one (so that a rank only loads
    what it analyses).  Returns the list of per-scan result lists, in the order of ``scans``; entries of other
    ranks are filled in when ``gather``, else left as None.  What travels is what analyze_sharded sends: one
    packed record per window (here with the window's centre and the number of its scan) and the candidates, in
    fixed-size all-gathers (RCCL on the GPUs) - after one gather of every rank's window count; each rank then
    builds the result dicts of all scans from the records, the same dicts its owner built.
    Without an initialised process group the one caller takes every scan."""
    from concurrent.futures import ThreadPoolExecutor

    from . import _ffi
    from ._ffi import DeviceScan

    world, rank = 1, 0
    dist = None
    try:
        import torch.distributed as dist_mod
        if dist_mod.is_available() and dist_mod.is_initialized():
            dist = dist_mod
            world, rank = dist.get_world_size(group), dist.get_rank(group)
    except ImportError:
        pass
    mine = scans_of_rank(len(scans), rank, world)
    # The upload thread below starts on device 0 whatever this thread has selected (the current device is a
    # property of the thread), and a scan can only be analysed on the device it was uploaded to: hand the
    # caller's device over.
    device = _ffi.current_device()

    def resident(k):
        _ffi.set_device(device)
        item = scans[k]
        pts = item() if callable(item) else item
        pts = np.ascontiguousarray(pts, dtype=np.float64)
        return analyzer._resident(pts)

    def results_of(z_centers, h):
        res = [r for r, _, warn in analyzer._results_of(z_centers, h) if warn is None]
        res.sort(key=lambda r: r["z_center"])
        return res

    exchange = dist is not None and gather and world > 1
    out = [None] * len(scans)
    packed = {}
    with ThreadPoolExecutor(1) as pool:
        nxt = pool.submit(resident, mine[0]) if mine else None
        for j, k in enumerate(mine):
            cur = nxt.result()
            nxt = pool.submit(resident, mine[j + 1]) if j + 1 < len(mine) else None
            try:
                z_centers = analyzer.calculate_z_centers(cur)
                h = analyzer._hybrid_windows(cur, z_centers) if len(z_centers) else None
            finally:
                if isinstance(cur, DeviceScan):
                    cur.close()
            out[k] = results_of(z_centers, h) if h is not None else []
            if exchange and h is not None:
                packed[k] = (z_centers, h)
            if progress_callback:
                progress_callback(int(100 * (j + 1) / len(mine)))
    if exchange:
        for k, (z_centers, h) in allgather_scan_records(packed, group).items():
            if out[k] is None:
                out[k] = results_of(z_centers, h)
        for k in range(len(scans)):
            if out[k] is None:                       # (a scan without a window: its owner sent nothing)
                out[k] = []
    return out
