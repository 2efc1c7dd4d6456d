"""One round of tests/fuzz_parity.py again, with the differences of the fused residual statistics printed
(python tools/exp_fuzz_case.py seed slice)."""
import os, sys
import numpy as np
import torch  # noqa: F401
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import fuzz_parity as Z
from pointcloud_amd import _ffi as F
from oracle import native as N
seed, s = int(sys.argv[1]), int(sys.argv[2])
g = np.random.default_rng(seed)
S = int(g.integers(3, 14))
sizes = [int(g.choice(Z.EDGE_SIZES)) if g.random() < 0.6 else int(g.integers(0, 9000)) for _ in range(S)]
kinds, sets = zip(*(Z.make_slice(g, n) for n in sizes))
x = np.concatenate([q[:, 0] for q in sets]); y = np.concatenate([q[:, 1] for q in sets])
off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
for mi in (100, 0):
    fu = Z.fused_dev(x, y, off, 64, -0.5, 0.5, mi)
    about = np.where(np.isfinite(fu["fit"]).all(axis=1)[:, None], fu["fit"], 1.0)
    chk = F.residuals_batch(x, y, off, about, bins=64, lo=-0.5, hi=0.5)
    print("max_iter", mi, kinds[s], sizes[s], "fit", fu["fit"][s].tolist(), "iters", fu["iters"][s].tolist())
    print("  hist equal", np.array_equal(fu["hist"][s], chk["hist"][s]), "diff at", np.flatnonzero(fu["hist"][s] != chk["hist"][s]).tolist(),
          (fu["hist"][s] - chk["hist"][s])[fu["hist"][s] != chk["hist"][s]].tolist())
    print("  argmax", fu["amm"][s, 0], chk["argmax"][s], "argmin", fu["amm"][s, 1], chk["argmin"][s])
    print("  rmax", repr(fu["rmm"][s, 0]), repr(chk["rmax"][s]), "rmin", repr(fu["rmm"][s, 1]), repr(chk["rmin"][s]))
    o0, o1 = off[s], off[s + 1]
    r = chk["radius"] if chk["radius"] is not None else None
    dr = chk["dr"][o0:o1]
    srt = np.sort(dr)
    print("  largest dr", srt[-3:].tolist(), "smallest", srt[:3].tolist())
    e = np.linspace(-0.5, 0.5, 65)
    t = (dr + 0.5) * 64
    print("  closest to an edge (bins):", np.min(np.abs(t - np.round(t))))
