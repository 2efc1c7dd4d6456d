"""Randomised comparison of the device replay of Qhull's construction (through pcfit_unique_hull_batch with
hull_start = PCFIT_HULL_START_QHULL: device kernel first, host replay for what it hands back) with the host
replay alone (pcfit_qhull_order_batch, itself checked against scipy by tests/sweep_qhull_order.py).  Needs a GPU.
By hand:  python tests/sweep_qhull_device.py [rounds] [seed]; tests/test_randomised.py runs a few fixed-seed
rounds of it under pytest."""
import os
import sys
import time

import numpy as np
import torch  # noqa: F401  (first, so that both use the HIP runtime torch ships)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.qhull2d import KINDS, make_case     # noqa: E402  (inputs only)
from pointcloud_amd import _ffi as F            # noqa: E402


def main(rounds=20, seed=0):
    gen = np.random.default_rng(seed)
    bad = open_ = slices = points = handed = quant = 0
    why_all = np.zeros(16, np.int64)
    t0 = time.time()
    # SWEEP_FORMS=1: every round in another form of the kernel (facet slots and threads; PCFIT_QHULL_FORM is read at
    # every call), the narrow ones that the launcher keeps for batches of thousands of slices among them; hulls
    # beyond a form's slots are handed back
    forms = np.random.default_rng(seed + 1)
    for r in range(rounds):
        if os.environ.get("SWEEP_FORMS"):
            os.environ["PCFIT_QHULL_FORM"] = str(forms.choice([96, 97, 98, 99, 113, 115, 128, 129, 130, 131, 145, 147, 160, 161, 162, 163, 224,
                                                               225, 226, 227, 480, 481, 482]))
        xs, ys, off = [], [], [0]
        for k in range(int(gen.integers(50, 600))):
            n = int(gen.choice([3, 4, 6, 10, 30, 200, 1500, 8192, 9000, 30000, 70000]))
            if gen.random() < 0.25:
                # a ring with coordinates rounded as in a text file: points exactly in line at the extremes
                th = gen.uniform(0, 2 * np.pi, n)
                rad = 10 + gen.normal(0, 0.005, n)
                xy = np.unique(np.round(np.stack([3 + rad * np.cos(th), -2 + rad * np.sin(th)], 1), int(gen.choice([2, 3, 3, 4]))),
                               axis=0)
                quant += 1
            else:
                xy = make_case(KINDS[int(gen.integers(len(KINDS)))], n, int(gen.integers(1 << 30)))
            if gen.random() < 0.2:                       # duplicates and another order: the library sorts and dedups
                xy = np.concatenate([xy, xy[gen.integers(0, len(xy), len(xy) // 3 + 1)]])
            elif gen.random() < 0.15:                    # every point several times over: copies of a set's farthest
                xy = np.repeat(xy, int(gen.integers(2, 6)), axis=0)      # point meet in different waves
            xy = xy[gen.permutation(len(xy))]
            xs.append(xy[:, 0]), ys.append(xy[:, 1]), off.append(off[-1] + len(xy))
        x, y, off = np.concatenate(xs), np.concatenate(ys), np.array(off, np.int64)
        why = np.zeros(16, np.int64)
        F.check(F.lib().pcfit_debug_qhull_reasons(F._p(why), 1))
        h = F.unique_hull_batch(x, y, off, hull_start="qhull")
        F.check(F.lib().pcfit_debug_qhull_reasons(F._p(why), 1))
        why_all += why
        host = F.qhull_order_batch(h["ux"], h["uy"], off, n_unique=h["n_unique"])
        for s in range(len(off) - 1):
            nh = int(h["n_hull"][s])
            if nh < 3 or host["start"][s] < 0:
                continue
            walk = h["hull_idx"][off[s]:off[s] + nh]
            hw = host["walk"][off[s]:off[s] + host["n_walk"][s]]
            same = np.array_equal(walk, hw)          # start and vertex set: Qhull's, merged vertices left out
            if not same and not host["sure"][s]:
                open_ += 1          # points exactly in line: the host replay's answer is a best effort itself
                if os.path.isdir("gpurun_out"):
                    nu = int(h["n_unique"][s])
                    np.save(f"gpurun_out/qhull_open_{seed}_{r}_{s}.npy",
                            np.stack([h["ux"][off[s]:off[s] + nu], h["uy"][off[s]:off[s] + nu]], 1))
                    print("open: round", r, "slice", s, "distinct points", nu, "device route", walk[:3], "host", hw[:3])
            elif not same:
                bad += 1
                print("round", r, "slice", s, "points", off[s + 1] - off[s], "device route", walk[:4], "host", hw[:4],
                      "sure", bool(host["sure"][s]))
        # the route of the analyzer (nobody asks for the distinct points: the replay reads the slices as they are,
        # unsorted and with their duplicates) against the route over the sorted distinct points
        raw = F.hybrid_batch(x, y, off, max_trials=64, want_unique=False, want_profile=False, order="qhull")
        srt = F.hybrid_batch(x, y, off, max_trials=64, hull_start="qhull", want_unique=True, want_profile=False)
        for key in ("status", "cand_offsets", "cand_x", "cand_y", "mask"):
            if not np.array_equal(raw[key], srt[key]):
                bad += 1
                co = srt["cand_offsets"]
                first = next((s for s in range(len(off) - 1)
                              if not np.array_equal(raw["cand_x"][co[s]:co[s + 1]], srt["cand_x"][co[s]:co[s + 1]])), None)
                print("round", r, "hybrid routes differ in", key, "first slice", first,
                      None if first is None else off[first + 1] - off[first])
                break
        slices += len(off) - 1
        points += int(off[-1])
        handed += int(why[:14].sum())
        if (r + 1) % 10 == 0:
            print(f"round {r + 1}/{rounds}: {slices + len(off) - 1} slices, {bad} differences", flush=True)
    print(f"{rounds} rounds, {slices} slices, {points} points, {time.time() - t0:.0f} s: {bad} differences "
          f"(and {open_} on slices the host replay is not sure of); {quant} slices with rounded coordinates; "
          f"handed back to the host {handed} ({why_all[:14].tolist()})")
    return bad, open_, slices


if __name__ == "__main__":
    sys.exit(1 if main(*(int(a) for a in sys.argv[1:3]))[0] else 0)
