"""Worker for the world_size > 1 tests (launched by torch.distributed.run)."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pointcloud_amd import sharding  # noqa: E402


def main():
    mode = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    ok = True
    if mode == "gather":
        gen = np.random.default_rng(123)                       # same stream on every rank
        for S in (0, 1, 2, 7, 40):
            counts = gen.integers(0, 5000, S)
            fit_all = gen.normal(size=(S, 3))
            hist_all = gen.integers(0, 100, (S, 16)).astype(np.int32)
            bounds = sharding.split_by_points(counts, world)
            s0, s1 = sharding.shard_range(bounds, rank)
            got = sharding.allgather_slice_arrays([torch.from_numpy(fit_all[s0:s1].copy()),
                                                   torch.from_numpy(hist_all[s0:s1].copy())], bounds)
            ok &= np.array_equal(got[0].numpy(), fit_all) and np.array_equal(got[1].numpy(), hist_all)
    elif mode == "records":
        # the packed per-window records and the two fixed-size gathers, on made-up results (no GPU)
        gen = np.random.default_rng(7)                         # same stream on every rank
        for S in (0, 1, 2, 9, 57):
            n0 = gen.integers(0, 9000, S)
            nc = gen.integers(0, 200, S) * (gen.random(S) < 0.8)
            co = np.concatenate([[0], np.cumsum(nc)]).astype(np.int64)
            full = dict(offsets=np.concatenate([[0], np.cumsum(n0)]).astype(np.int64),
                        status=gen.integers(0, 4, S).astype(np.int32), cand_offsets=co,
                        cand_x=gen.normal(size=co[-1]), cand_y=gen.normal(size=co[-1]),
                        argmax=gen.integers(0, 100, S), argmin=gen.integers(0, 100, S),
                        fit=gen.normal(size=(S, 3)), ov=gen.normal(size=(S, 5)))
            if S:
                full["fit"][0, 1] = np.nan                     # a failed window travels as NaN
            bounds = sharding.split_by_points(n0, world)
            s0, s1 = sharding.shard_range(bounds, rank)
            mine = dict(offsets=full["offsets"][s0:s1 + 1] - full["offsets"][s0], status=full["status"][s0:s1],
                        cand_offsets=co[s0:s1 + 1] - co[s0], cand_x=full["cand_x"][co[s0]:co[s1]],
                        cand_y=full["cand_y"][co[s0]:co[s1]], argmax=full["argmax"][s0:s1],
                        argmin=full["argmin"][s0:s1], fit=full["fit"][s0:s1], ov=full["ov"][s0:s1])
            rec, cand = sharding.pack_window_records(mine)
            recs, cands = sharding.allgather_window_records(rec, cand, bounds)
            got = sharding.unpack_window_records(recs, cands)
            for k, v in full.items():
                ok &= np.array_equal(got[k], v, equal_nan=True) and got[k].dtype == np.asarray(v).dtype
    elif mode == "scan_records":
        # a batch of whole scans dealt to the ranks round-robin: what every rank learns of the others' scans through
        # allgather_scan_records is what their owners computed (made-up results, no GPU); scans without windows
        # travel as nothing, a world larger than the batch leaves ranks without a scan
        gen = np.random.default_rng(11)                        # same stream on every rank
        for n_scans in (0, 1, 2, 5, 7):
            full = {}
            for k in range(n_scans):
                S = int(gen.integers(0, 30)) if k != 3 else 0
                n0 = gen.integers(0, 9000, S)
                nc = gen.integers(0, 200, S) * (gen.random(S) < 0.8)
                co = np.concatenate([[0], np.cumsum(nc)]).astype(np.int64)
                h = dict(offsets=np.concatenate([[0], np.cumsum(n0)]).astype(np.int64),
                         status=gen.integers(0, 4, S).astype(np.int32), cand_offsets=co,
                         cand_x=gen.normal(size=co[-1]), cand_y=gen.normal(size=co[-1]),
                         argmax=gen.integers(0, 100, S), argmin=gen.integers(0, 100, S),
                         fit=gen.normal(size=(S, 3)), ov=gen.normal(size=(S, 5)))
                full[k] = (np.sort(gen.normal(size=S)) * 100.0, h)
            mine = {k: full[k] for k in sharding.scans_of_rank(n_scans, rank, world) if len(full[k][0])}
            got = sharding.allgather_scan_records(mine)
            ok &= sorted(got) == [k for k in range(n_scans) if len(full[k][0])]
            for k, (zc, h) in got.items():
                ok &= np.array_equal(zc, full[k][0])
                for key, v in full[k][1].items():
                    ok &= np.array_equal(h[key], v, equal_nan=True) and h[key].dtype == np.asarray(v).dtype
    elif mode in ("analyze_gpu", "analyze_gpu_ranged"):
        from oracle import ref_np as R
        from pointcloud_amd import Config, CylinderAnalyzer, DeviceScan
        pts = R.make_cylinder(60_000, 12, seed=17, outlier_frac=0.02)
        # points exactly on window edges, shared ones included: they belong to two windows, and where the
        # two windows belong to two ranks both ranks must have uploaded them
        for k, zc in enumerate(np.arange(0.0, 24.5, 2.0)):
            pts[100 + 5 * k:105 + 5 * k, 2] = zc
        if mode == "analyze_gpu":
            az = CylinderAnalyzer(Config(z_step=2.0, min_points_per_slice=500))
            want_windows = 12
        else:
            az = CylinderAnalyzer(Config(z_step=2.0, min_points_per_slice=500, z_min=3.0, z_max=20.0))
            want_windows = 8
        res = sharding.analyze_sharded(az, pts)
        single = az.analyze_with_progress(pts)
        ok &= len(res) == len(single) == want_windows
        ok &= res == single                                    # every key of every dict, floats bit for bit
        if mode == "analyze_gpu_ranged":
            # the last window's upper edge lies a rounding above z_max (np.arange's slack): rows between the two
            # are dropped by the configured range before any window sees them, on one rank and on several
            fine = CylinderAnalyzer(Config(z_step=0.1, min_points_per_slice=20, z_min=0.0, z_max=1.0))
            top = fine.calculate_z_centers(pts)[-1] + 0.05
            ok &= top > 1.0
            p2 = pts.copy()
            p2[:, 2] = np.random.default_rng(5).uniform(0.0, 1.0, len(p2))
            p2[:400, 2] = np.nextafter(1.0, 2.0)               # above z_max, at or below the last window's edge
            p2[400:800, 2] = 1.0
            one = fine.analyze_with_progress(p2)
            ok &= len(one) == 10 and sharding.analyze_sharded(fine, p2) == one
            ok &= one[-1]["n_points_slice"] == int(np.count_nonzero((p2[:, 2] >= one[-1]["z_low"]) & (p2[:, 2] <= 1.0)))
        if mode == "analyze_gpu":
            with DeviceScan(pts) as scan:                      # a scan that is already on the rank's GPU, whole
                ok &= sharding.analyze_sharded(az, scan) == single
            # a rank whose range holds no window, and windows that are all too sparse
            few = CylinderAnalyzer(Config(z_step=12.0, min_points_per_slice=500))
            ok &= sharding.analyze_sharded(few, pts) == few.analyze_with_progress(pts)
            sparse = CylinderAnalyzer(Config(z_step=2.0, min_points_per_slice=10**6))
            ok &= sharding.analyze_sharded(sparse, pts) == []
    elif mode == "scans_gpu":
        from oracle import ref_np as R
        from pointcloud_amd import Config, CylinderAnalyzer
        az = CylinderAnalyzer(Config(z_step=2.0, min_points_per_slice=500))
        loaded = []

        def scan(k):
            def load():
                loaded.append(k)
                return R.make_cylinder(40_000, 8, seed=200 + k, outlier_frac=0.01)
            return load
        res = sharding.analyze_scans_sharded(az, [scan(k) for k in range(5)])
        ok &= loaded == sharding.scans_of_rank(5, rank, world)          # a rank loads only its own scans
        ok &= len(res) == 5 and all(r is not None and len(r) == 8 for r in res)
        # every rank holds every scan's results, those of the others rebuilt from the packed records: the dicts
        # a single process computes, floats bit for bit
        for k in (rank % 5, (rank + 1) % 5, 4):
            single = az.analyze_with_progress(R.make_cylinder(40_000, 8, seed=200 + k, outlier_frac=0.01))
            ok &= res[k] == single
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(json.dumps({"ok": bool(flag.item()), "world": world, "mode": mode}))
    dist.destroy_process_group()
    sys.exit(0 if flag.item() else 1)


if __name__ == "__main__":
    main()
