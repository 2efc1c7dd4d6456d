"""Slice sharding: the partition is a pure function; the gather runs in real
world_size 2 and 3 process groups over gloo (CPU)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from pointcloud_amd import sharding


def test_split_is_contiguous_complete_and_balanced():
    gen = np.random.default_rng(0)
    for world in (1, 2, 3, 8):
        for S in (0, 1, 5, 8, 200, 2000):
            counts = gen.integers(0, 60_000, S)
            b = sharding.split_by_points(counts, world)
            assert b[0] == 0 and b[-1] == S and np.all(np.diff(b) >= 0)
            if S >= 50 * world:
                load = np.array([counts[b[r]:b[r + 1]].sum() for r in range(world)])
                assert load.max() - load.min() <= 2 * counts.max()
    # uniform slices split evenly; a slice holding everything keeps its neighbours legal
    assert list(sharding.split_by_points(np.full(16, 10), 8)) == list(range(0, 17, 2))
    b = sharding.split_by_points(np.array([0, 0, 1000, 0]), 2)
    assert b[0] == 0 and b[-1] == 4 and np.all(np.diff(b) >= 0)
    assert list(sharding.split_by_points(np.zeros(6, int), 3)) == [0, 2, 4, 6]


def run_workers(world, mode, timeout=300):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29500 + world + (7 if mode != "gather" else 0)),
           os.path.join(ROOT, "tests", "dist_worker.py"), mode]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


@pytest.mark.parametrize("world", [2, 3])
def test_allgather_of_slice_arrays_over_gloo(world):
    r = run_workers(world, "gather")
    assert r["ok"] and r["world"] == world


@pytest.mark.parametrize("world", [2, 3])
def test_window_records_travel_packed_over_gloo(world):
    r = run_workers(world, "records")
    assert r["ok"] and r["world"] == world


@pytest.mark.parametrize("world", [2, 3])
def test_scan_batch_records_travel_packed_over_gloo(world):
    r = run_workers(world, "scan_records")
    assert r["ok"] and r["world"] == world


def test_host_passes_over_the_z_column():
    """numpy's min / max and the closed-window populations (shared edges counted twice, NaN nowhere), computed by
    the library's host threads: host code only, no GPU needed."""
    from oracle import ref_np as R
    from pointcloud_amd import _ffi as F
    pts = R.make_cylinder(300_000, 40, seed=3, outlier_frac=0.01)
    zc = np.arange(1.0, 80.0, 2.0)
    lo, hi = zc - 1.0, zc + 1.0
    pts[50:90, 2] = np.repeat(hi[:20], 2)                      # on shared edges
    pts[90:100, 2] = np.nextafter(hi[:10], np.inf)             # one ulp past an edge: the next window only
    assert F.host_z_range(pts) == (pts[:, 2].min(), pts[:, 2].max())
    want = np.array([np.count_nonzero((pts[:, 2] >= a) & (pts[:, 2] <= b)) for a, b in zip(lo, hi)])
    assert np.array_equal(F.host_window_counts(pts, lo, hi), want)
    assert np.array_equal(F.host_window_counts(pts, lo[3:4], hi[3:4], n_threads=1), want[3:4])
    gaps = F.host_window_counts(pts, zc - 0.25, zc + 0.25, n_threads=3)
    assert np.array_equal(gaps, [np.count_nonzero(np.abs(pts[:, 2] - c) <= 0.25) for c in zc])
    pts[7, 2] = np.nan
    assert all(np.isnan(F.host_z_range(pts)))
    want = np.array([np.count_nonzero((pts[:, 2] >= a) & (pts[:, 2] <= b)) for a, b in zip(lo, hi)])
    assert np.array_equal(F.host_window_counts(pts, lo, hi), want)
    assert F.host_z_range(np.zeros((0, 3))) != F.host_z_range(np.zeros((0, 3)))      # NaN, NaN


@pytest.mark.gpu
@pytest.mark.parametrize("world,mode", [(2, "analyze_gpu"), (3, "analyze_gpu"), (2, "analyze_gpu_ranged")])
def test_sharded_analyzer_ranks_share_one_gpu(world, mode):
    # the ranks share the one GPU for compute, upload only their own z range and exchange packed records over gloo
    r = run_workers(world, mode)
    assert r["ok"]


@pytest.mark.gpu
def test_upload_of_a_z_range_keeps_rows_and_order():
    from oracle import ref_np as R
    from pointcloud_amd import _ffi as F
    pts = R.make_cylinder(1_500_000, 30, seed=8)               # several staging chunks
    lo, hi = 11.0, 25.0
    pts[1000:1010, 2] = lo
    pts[2000:2010, 2] = hi
    keep = (pts[:, 2] >= lo) & (pts[:, 2] <= hi)
    with F.DeviceScan(pts, z_range=(lo, hi)) as part:
        assert len(part) == int(keep.sum())
        off, x, y, perm = F.bin_slices(part, [lo], [hi])
        assert off[1] == len(part) and np.array_equal(perm, np.arange(len(part)))
        assert np.array_equal(x, pts[keep, 0]) and np.array_equal(y, pts[keep, 1])
    with F.DeviceScan(pts, z_range=(1000.0, 2000.0)) as none:
        assert len(none) == 0


def test_scans_go_round_robin():
    assert sharding.scans_of_rank(64, 3, 8) == list(range(3, 64, 8))
    assert sharding.scans_of_rank(2, 5, 8) == [] and sharding.scans_of_rank(0, 0, 1) == []
    assert sorted(k for r in range(3) for k in sharding.scans_of_rank(10, r, 3)) == list(range(10))


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_batch_of_scans_ranks_share_one_gpu(world):
    # five scans over two ranks (3 + 2) or three (2 + 2 + 1), uploads overlapped with the analyses, the packed
    # window records exchanged over gloo, every rank's copy of every scan's results checked
    r = run_workers(world, "scans_gpu")
    assert r["ok"] and r["world"] == world


@pytest.mark.gpu
def test_scan_batches_run_on_the_callers_device():
    """The upload thread of analyze_scans_sharded must work on the device the calling thread selected (a new
    thread starts on device 0).  With one GPU the device is 0 either way: there the hand-over itself is checked;
    with more the main thread selects GPU 1."""
    import ctypes as C
    import threading

    from oracle import ref_np as R
    from pointcloud_amd import Config, CylinderAnalyzer
    from pointcloud_amd import _ffi as F
    n = C.c_int(0)
    F.check(F.lib().pcfit_device_count(C.byref(n)))
    dev = 1 if n.value > 1 else 0
    F.set_device(dev)
    try:
        assert F.current_device() == dev
        seen = []
        t = threading.Thread(target=lambda: seen.append(F.current_device()))
        t.start()
        t.join()
        assert seen == [0]                                     # the rule the hand-over exists for
        az = CylinderAnalyzer(Config(z_step=2.0, min_points_per_slice=500))
        scans = [R.make_cylinder(30_000, 6, seed=300 + k) for k in range(3)]
        res = sharding.analyze_scans_sharded(az, scans)
        assert [len(r) for r in res] == [6, 6, 6]
        with F.DeviceScan(scans[0]) as sc:
            assert sc.device == dev
            assert res[0] == az.analyze_with_progress(sc)
    finally:
        F.set_device(0)
