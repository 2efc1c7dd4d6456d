import os
import sys

import numpy as np
import pytest
import torch  # noqa: F401  (first, so that the process runs on ONE HIP runtime: the one torch ships)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def ulp_diff(a, b):
    """Distance in units in the last place between float64 arrays."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    ia = a.view(np.int64).copy()
    ib = b.view(np.int64).copy()
    ia[ia < 0] = np.int64(-2**63) - ia[ia < 0]
    ib[ib < 0] = np.int64(-2**63) - ib[ib < 0]
    return np.abs(ia - ib)


def same_cycle(a, b):
    """True when sequence b is a cyclic rotation of sequence a."""
    a = list(a)
    b = list(b)
    if len(a) != len(b):
        return False
    if not a:
        return True
    if b[0] not in a:
        return False
    k = a.index(b[0])
    return a[k:] + a[:k] == b


def year_fixture():
    """The window state the GUI's year export ran on (results with boundary points) and what it wrote."""
    g = load_golden("year_export")
    off = g["boundary_offsets"]
    results = []
    for s, (z, cx, cy, r, ov) in enumerate(g["scalars"]):
        b = g["boundary_xy"][off[s]:off[s + 1]]
        results.append(dict(z_center=float(z), cx=float(cx), cy=float(cy), R=float(r), ovality_pct=float(ov),
                            boundary_points=b.tolist()))
    return g, results
