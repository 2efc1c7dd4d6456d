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
    elif mode == "analyze_gpu":
        from oracle import ref_np as R
        from pointcloud_amd import Config, CylinderAnalyzer
        pts = R.make_cylinder(60_000, 12, seed=17, outlier_frac=0.0)
        az = CylinderAnalyzer(Config(z_step=2.0, min_points_per_slice=500))
        res = sharding.analyze_sharded(az, pts)
        if rank == 0:
            single = az.analyze_with_progress(pts)
            ok &= len(res) == len(single) == 12
            for a, b in zip(res, single):
                ok &= a["z_center"] == b["z_center"] and a["cx"] == b["cx"] and a["R"] == b["R"]
                ok &= a["boundary_points"] == b["boundary_points"] and a["n_points_slice"] == b["n_points_slice"]
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
        if rank == 0:
            for k in (0, 3, 4):
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
