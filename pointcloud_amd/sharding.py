"""Slice sharding over the GPUs of one node.

Slices are independent (each window reads only its own points), so the path
shards with no exchange of points: every rank takes a contiguous range of
windows, balanced by point count, and only per-slice results travel - fitted
parameters and residual histograms, all-gathered with torch.distributed
(backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in CPU tests).
"""
from __future__ import annotations

import numpy as np


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


def analyze_sharded(analyzer, points, group=None, progress_callback=None):
    """CylinderAnalyzer.analyze_with_progress with the windows sharded over the
    ranks of ``group``: every rank bins and fits its own range of windows on its
    GPU, then the per-slice result dicts are exchanged so that every rank returns
    the full, z-ordered list.  ``points`` is the whole scan on every rank, as an
    (N, 3) array or a DeviceScan already on the rank's GPU."""
    import torch.distributed as dist

    from ._ffi import DeviceScan

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dz = analyzer.config.z_step / 2.0
    ranged = analyzer.z_min is not None and analyzer.z_max is not None
    if isinstance(points, DeviceScan) or not ranged:
        # The scan goes to this rank's GPU once, whole: the z range, the window populations (the count pass of
        # the binning) and the analysis of the rank's own windows all read that copy, and points of other ranks'
        # windows are simply never binned. No pass over the points on the host.
        own = not isinstance(points, DeviceScan)
        if own:
            points = np.ascontiguousarray(points, dtype=np.float64)
            if points.ndim != 2 or points.shape[1] != 3:
                raise ValueError("Points must be Nx3 array")
        elif ranged:
            raise ValueError("a DeviceScan cannot be combined with Config.z_min / z_max: pass the array")
        scan = DeviceScan(points) if own and len(points) else points
        try:
            z_centers = analyzer.calculate_z_centers(scan)
            counts = (scan.window_counts(z_centers - dz, z_centers + dz) if len(z_centers) and len(scan)
                      else np.zeros(len(z_centers), np.int64))
            bounds = split_by_points(counts, world)
            s0, s1 = shard_range(bounds, rank)
            mine = []
            if s1 > s0:
                for res, _, warn in analyzer._analyze_windows(scan, z_centers[s0:s1]):
                    if warn is None:
                        mine.append(res)
        finally:
            if own and isinstance(scan, DeviceScan):
                scan.close()
    else:
        # a configured z range filters the points first (analyzer.py:190-196), which needs the array
        points = np.ascontiguousarray(points, dtype=np.float64)
        if points.ndim != 2 or points.shape[1] != 3:
            raise ValueError("Points must be Nx3 array")
        points = points[(points[:, 2] >= analyzer.z_min) & (points[:, 2] <= analyzer.z_max)]
        if len(points) == 0:
            raise ValueError(f"No points in specified Z range [{analyzer.z_min}, {analyzer.z_max}]")
        z_centers = analyzer.calculate_z_centers(points)
        # balance by the window populations (one histogram of z on the host)
        edges = np.concatenate([z_centers - dz, z_centers[-1:] + dz]) if len(z_centers) else np.zeros(1)
        counts = np.histogram(points[:, 2], bins=edges)[0] if len(z_centers) else np.zeros(0, np.int64)
        bounds = split_by_points(counts, world)
        s0, s1 = shard_range(bounds, rank)
        mine = []
        if s1 > s0:
            lo, hi = z_centers[s0] - dz, z_centers[s1 - 1] + dz
            sel = points[(points[:, 2] >= lo) & (points[:, 2] <= hi)]
            for res, _, warn in analyzer._analyze_windows(np.ascontiguousarray(sel), z_centers[s0:s1]):
                if warn is None:
                    mine.append(res)
    everyone = [None] * world
    dist.all_gather_object(everyone, mine, group=group)
    results = [r for part in everyone for r in part]
    results.sort(key=lambda r: r["z_center"])
    if progress_callback:
        progress_callback(100)
    return results


def scans_of_rank(n_scans: int, rank: int, world: int):
    """Whole scans go to the ranks round-robin: rank r analyses scans r, r + world, ..."""
    return list(range(rank, n_scans, world))


def analyze_scans_sharded(analyzer, scans, group=None, progress_callback=None, gather=True):
    """A batch of independent scans over the ranks of ``group``: whole scans round-robin to the ranks (no cuts
    through a scan, no exchange of points), each rank running its scans one after the other with the upload of
    the next one (a second host thread making the DeviceScan) overlapped with the analysis of the current one.
    ``scans``: a sequence whose items are (N, 3) arrays or callables returning one (so that a rank only loads
    what it analyses).  Returns the list of per-scan result lists, in the order of ``scans``; entries of other
    ranks are filled in when ``gather`` (all_gather_object of the result dicts), else left as None.
    Without an initialised process group the one caller takes every scan."""
    from concurrent.futures import ThreadPoolExecutor

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

    def resident(k):
        item = scans[k]
        pts = item() if callable(item) else item
        pts = np.ascontiguousarray(pts, dtype=np.float64)
        if pts.ndim != 2 or pts.shape[1] != 3:
            raise ValueError("Points must be Nx3 array")
        return DeviceScan(pts)

    out = [None] * len(scans)
    with ThreadPoolExecutor(1) as pool:
        nxt = pool.submit(resident, mine[0]) if mine else None
        for j, k in enumerate(mine):
            cur = nxt.result()
            nxt = pool.submit(resident, mine[j + 1]) if j + 1 < len(mine) else None
            try:
                out[k] = analyzer.analyze_with_progress(cur)
            finally:
                cur.close()
            if progress_callback:
                progress_callback(int(100 * (j + 1) / len(mine)))
    if dist is not None and gather and world > 1:
        everyone = [None] * world
        dist.all_gather_object(everyone, {k: out[k] for k in mine}, group=group)
        for part in everyone:
            for k, res in part.items():
                out[k] = res
    return out
