"""One hybrid round of tests/fuzz_parity.py by seed, with what differs, the whole batch as the sweep calls it (needs a
GPU): python tools/exp_fuzz_hybrid_case.py seed"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa
import fuzz_parity as Z
from pointcloud_amd import _ffi as F
from oracle import ref_np as R

seed = int(sys.argv[1])
g = np.random.default_rng(seed)
S = int(g.integers(2, 6))
sizes = [int(g.choice([0, 2, 3, 5, 40, 300, 1024, 1025, 2500, 6000, 8192])) for _ in range(S)]
kinds, sets = zip(*(Z.make_slice(g, n) for n in sizes))
x = np.concatenate([s[:, 0] for s in sets]) if sum(sizes) else np.zeros(0)
y = np.concatenate([s[:, 1] for s in sets]) if sum(sizes) else np.zeros(0)
off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
use_hull = bool(g.random() < 0.7)
trials = int(g.choice([50, 200]))
want_unique = bool(g.random() < 0.4)
qhull = bool(use_hull and g.random() < 0.5)
print("kinds", kinds, "sizes", sizes, "use_hull", use_hull, "want_unique", want_unique, "qhull", qhull)
refs = [R.hybrid_fit(np.column_stack([xy, np.zeros(len(xy))]), 0.0, 1.0, max_trials=trials, use_hull_vertices_only=use_hull,
                     hull_start="qhull" if qhull else None) if len(xy) else None for xy in sets]
for env in [{}] * int(os.environ.get("REPEAT", "1")) + [{"PCFIT_QHULL_ON_HOST": "1"}]:
    for k in ("PCFIT_QHULL_ON_HOST", "PCFIT_PINNED_SCRATCH"):
        os.environ.pop(k, None)
    os.environ.update(env)
    why = np.zeros(16, np.int64)
    F.check(F.lib().pcfit_debug_qhull_reasons(F._p(why), 1))
    h = F.hybrid_batch(x, y, off, max_trials=trials, use_hull=use_hull, want_profile=False, want_unique=want_unique,
                       order="qhull" if qhull else "canonical")
    F.check(F.lib().pcfit_debug_qhull_reasons(F._p(why), 1))
    out = []
    for s, ref in enumerate(refs):
        if ref is None or h["status"][s] != 0:
            out.append(None)
            continue
        c0, c1 = h["cand_offsets"][s], h["cand_offsets"][s + 1]
        c = np.stack([h["cand_x"][c0:c1], h["cand_y"][c0:c1]], 1)
        same = np.array_equal(c, ref["candidates_xy"])
        out.append(same)
        if not same:
            rot = [k for k in range(len(c)) if len(c) == len(ref["candidates_xy"]) and np.array_equal(np.roll(c, -k, 0), ref["candidates_xy"])]
            print("   slice", s, "library", c[:3].tolist(), "oracle", ref["candidates_xy"][:3].tolist(), "n", len(c), len(ref["candidates_xy"]), "a rotation by", rot)
    print(env, "same as the oracle per slice:", out, "reasons", why[:14].tolist())
