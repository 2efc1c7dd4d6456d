"""First GPU check of kasa / lm / ransac / residual / angle-max kernels against the oracles."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ctypes as C
from pointcloud_amd import _ffi as F
from oracle import native as N, ref_np as R

# tolerant loader: only the symbols present
L = C.CDLL(F.SO_PATH)
for name, (res, args) in F.SIGNATURES.items():
    if hasattr(L, name):
        fn = getattr(L, name); fn.restype = res; fn.argtypes = args
F._lib = L

def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b)) / np.maximum(1.0, np.abs(b))))

x, y, off = R.make_slices(64, 8192, seed=1, jitter=0.05)
t0 = time.time(); k, st = F.kasa_batch(x, y, off); print("kasa call", time.time() - t0)
ko = np.array([N.kasa(x[off[s]:off[s+1]], y[off[s]:off[s+1]])[1] for s in range(64)])
print("kasa rel", rel(k, ko), st.max())
out, it = F.lm_batch(x, y, off, k)
lo = np.array([N.lm(x[off[s]:off[s+1]], y[off[s]:off[s+1]], ko[s])[0] for s in range(64)])
print("lm rel", rel(out, lo), it[:4].tolist())

# ragged + outliers
x2, y2, _ = R.make_slices(1, 40000, seed=2, outlier_frac=0.3)
cuts = np.sort(np.random.default_rng(0).integers(0, 40000, 30)); off2 = np.concatenate([[0], cuts, [40000]]).astype(np.int64)
off2[5] = off2[4]  # empty slice
off2[7] = off2[6] + 2 if off2[6] + 2 <= off2[8] else off2[7]
k2, st2 = F.kasa_batch(x2, y2, off2)
for s in range(len(off2) - 1):
    n = off2[s+1] - off2[s]
    so, ko2 = N.kasa(x2[off2[s]:off2[s+1]], y2[off2[s]:off2[s+1]])
    assert so == st2[s], (s, n, so, st2[s])
    if so == 0: assert rel(k2[s], ko2) < 1e-10, (s, n, k2[s], ko2)
print("ragged kasa ok; statuses", st2.tolist())

r = F.ransac_batch(x2, y2, off2, max_trials=256)
bad = 0
for s in range(len(off2) - 1):
    xy = np.stack([x2[off2[s]:off2[s+1]], y2[off2[s]:off2[s+1]]], 1)
    o = N.ransac(xy, trials=256)
    if o["status"] != r["status"][s] or o["score"] != r["score"][s] or o["best_trial"] != r["best_trial"][s]:
        bad += 1; print("mismatch", s, len(xy), o["status"], r["status"][s], o["score"], r["score"][s], o["best_trial"], r["best_trial"][s])
    elif o["status"] == 0:
        assert np.array_equal(o["mask"], r["mask"][off2[s]:off2[s+1]]), s
        assert np.array_equal(o["model"], r["model"][s]), (s, o["model"], r["model"][s])
print("ransac mismatches", bad)

x3, y3, off3 = R.make_slices(8, 8192, seed=3, outlier_frac=0.3)
t0 = time.time(); r3 = F.ransac_batch(x3, y3, off3, max_trials=1024); print("ransac call", time.time() - t0)
for s in range(8):
    xy = np.stack([x3[off3[s]:off3[s+1]], y3[off3[s]:off3[s+1]]], 1)
    o = N.ransac(xy, trials=1024)
    assert o["score"] == r3["score"][s] and o["best_trial"] == r3["best_trial"][s], (s, o["score"], r3["score"][s])
    assert np.array_equal(o["mask"], r3["mask"][off3[s]:off3[s+1]]) and np.array_equal(o["model"], r3["model"][s])
print("ransac 8x8192x1024 bit-exact")

fit = np.tile([3.01, -1.98, 10.02], (len(off2) - 1, 1))
res = F.residuals_batch(x2, y2, off2, fit, bins=256, lo=-0.5, hi=0.5)
sel = F.angle_max_batch(x2, y2, off2, fit, bins=720)
mu = 0
for s in range(len(off2) - 1):
    xy = np.stack([x2[off2[s]:off2[s+1]], y2[off2[s]:off2[s+1]]], 1)
    o = N.residuals(xy, *fit[s], bins=256, lo=-0.5, hi=0.5)
    sl = slice(off2[s], off2[s+1])
    assert np.array_equal(o["dr"], res["dr"][sl]), s
    mu = max(mu, int(np.max(np.abs(o["theta"].view(np.int64) - res["theta"][sl].view(np.int64)))) if len(xy) else 0)
    assert np.array_equal(o["hist"], res["hist"][s]), s
    assert o["argmax"] == res["argmax"][s] and o["argmin"] == res["argmin"][s], s
    if len(xy): assert o["rmax"] == res["rmax"][s] and o["rmin"] == res["rmin"][s]
    so = N.angle_max(xy, fit[s][0], fit[s][1], 720)
    if not np.array_equal(so, sel[s]): print("angle_max differs", s, np.count_nonzero(so != sel[s]))
print("residuals ok; theta max ulp diff", mu)
