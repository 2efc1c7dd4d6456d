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


@pytest.mark.gpu
def test_sharded_analyzer_two_ranks_one_gpu():
    # two ranks share the one GPU for compute and exchange results over gloo
    r = run_workers(2, "analyze_gpu")
    assert r["ok"]


def test_scans_go_round_robin():
    assert sharding.scans_of_rank(64, 3, 8) == list(range(3, 64, 8))
    assert sharding.scans_of_rank(2, 5, 8) == [] and sharding.scans_of_rank(0, 0, 1) == []
    assert sorted(k for r in range(3) for k in sharding.scans_of_rank(10, r, 3)) == list(range(10))


@pytest.mark.gpu
def test_batch_of_scans_two_ranks_one_gpu():
    # five scans over two ranks (3 + 2), uploads overlapped with the analyses, results exchanged over gloo
    r = run_workers(2, "scans_gpu")
    assert r["ok"]
