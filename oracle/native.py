"""ctypes binding of oracle/pcfit_oracle.c (TEST INFRASTRUCTURE ONLY).

``build()`` compiles the C file with gcc into oracle/_build; ``lib()`` loads it.
The wrappers take and return numpy arrays and mirror the C signatures.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "libpcfit_oracle.so")
_lib = None

f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "pcfit_oracle.c")
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-s"] + (["-B"] if force else []))
    return SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            build()
        L = C.CDLL(SO)
        L.orc_seed_pcg64.argtypes = [C.c_uint64, u64p]
        L.orc_choice3_table.argtypes = [u64p, C.c_int64, C.c_int32, i32p]
        L.orc_circle_3pt.argtypes = [f64p, f64p]
        L.orc_circle_3pt.restype = C.c_int
        L.orc_ransac.argtypes = [f64p, f64p, C.c_int64, C.c_double, C.c_int32, C.c_double, C.c_void_p, u64p,
                                 f64p, C.POINTER(C.c_int32), u8p, C.POINTER(C.c_int32)]
        L.orc_ransac.restype = C.c_int
        L.orc_kasa.argtypes = [f64p, f64p, C.c_int64, f64p]
        L.orc_kasa.restype = C.c_int
        L.orc_lm.argtypes = [f64p, f64p, C.c_int64, f64p, C.c_double, C.c_int32, C.c_double, C.c_double, C.c_double,
                             f64p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.orc_lm.restype = C.c_int
        L.orc_bin_slices.argtypes = [C.c_void_p, C.c_int64, C.c_int64, f64p, f64p, C.c_int64, i64p, i64p, C.c_int64]
        L.orc_bin_slices.restype = C.c_int64
        L.orc_sort_unique.argtypes = [f64p, f64p, C.c_int64, f64p, f64p]
        L.orc_sort_unique.restype = C.c_int64
        L.orc_has_2d_variation.argtypes = [f64p, f64p, C.c_int64, C.c_double]
        L.orc_has_2d_variation.restype = C.c_int
        L.orc_hull.argtypes = [f64p, f64p, C.c_int64, i64p]
        L.orc_hull.restype = C.c_int64
        L.orc_polar.argtypes = [f64p, f64p, C.c_int64, C.c_double, C.c_double, f64p, f64p]
        L.orc_angle_max.argtypes = [f64p, f64p, C.c_int64, C.c_double, C.c_double, C.c_int32, i64p]
        L.orc_residuals.argtypes = [f64p, f64p, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p,
                                    C.c_int32, f64p, i64p, f64p, i64p]
        L.orc_ovality.argtypes = [f64p, f64p, C.c_int64, C.c_double, C.c_double, C.c_double, f64p, C.c_void_p]
        L.orc_fit_residual_batch.argtypes = [f64p, f64p, i64p, C.c_int64, C.c_int32, f64p, f64p, i64p, f64p, i64p]
        _lib = L
    return _lib


def _c(a, dt=np.float64):
    return np.ascontiguousarray(a, dtype=dt)


def seed_state(seed: int = 42) -> np.ndarray:
    out = np.zeros(4, np.uint64)
    lib().orc_seed_pcg64(C.c_uint64(seed), out)
    return out


def choice3_table(n: int, trials: int, seed: int = 42) -> np.ndarray:
    out = np.zeros((trials, 3), np.int32)
    lib().orc_choice3_table(seed_state(seed), n, trials, out)
    return out


def circle_3pt(p1, p2, p3):
    out = np.zeros(3)
    ok = lib().orc_circle_3pt(_c(np.concatenate([p1, p2, p3])), out)
    return tuple(out) if ok else None


def ransac(xy, tol=0.5, trials=4000, min_frac=0.1, r_range=None, seed=42):
    x, y = _c(xy[:, 0]), _c(xy[:, 1])
    n = len(x)
    model = np.zeros(3)
    mask = np.zeros(max(n, 1), np.uint8)
    score, best = C.c_int32(), C.c_int32()
    rr = None if r_range is None else _c(r_range)
    st = lib().orc_ransac(x, y, n, tol, trials, min_frac, None if rr is None else rr.ctypes.data, seed_state(seed),
                          model, C.byref(score), mask, C.byref(best))
    return dict(status=st, score=score.value, best_trial=best.value, model=model, mask=mask[:n].astype(bool))


def kasa(x, y):
    out = np.full(3, np.nan)
    st = lib().orc_kasa(_c(x), _c(y), len(x), out)
    return st, out


def lm(x, y, init, delta=1.0, max_iter=100, lam0=1e-2, tol_update=1e-10, tol_cost=1e-12):
    out = np.zeros(3)
    it, acc = C.c_int32(), C.c_int32()
    lib().orc_lm(_c(x), _c(y), len(x), _c(init), delta, max_iter, lam0, tol_update, tol_cost, out,
                 C.byref(it), C.byref(acc))
    return out, it.value, acc.value


def bin_slices(points_or_z, lo, hi):
    a = np.asarray(points_or_z)
    if a.ndim == 2:
        base = np.ascontiguousarray(a, dtype=np.float64)
        ptr, stride, n = base.ctypes.data + 16, 3, base.shape[0]
    else:
        base = _c(a)
        ptr, stride, n = base.ctypes.data, 1, base.shape[0]
    lo, hi = _c(lo), _c(hi)
    S = len(lo)
    offsets = np.zeros(S + 1, np.int64)
    cap = 2 * n + 1
    perm = np.zeros(cap, np.int64)
    tot = lib().orc_bin_slices(ptr, n, stride, lo, hi, S, offsets, perm, cap)
    return offsets, perm[:tot]


def sort_unique(xy):
    x, y = _c(xy[:, 0]), _c(xy[:, 1])
    xo, yo = np.zeros_like(x), np.zeros_like(y)
    m = lib().orc_sort_unique(x, y, len(x), xo, yo)
    return np.stack([xo[:m], yo[:m]], axis=1)


def has_2d_variation(xy, eps=1e-12):
    return bool(lib().orc_has_2d_variation(_c(xy[:, 0]), _c(xy[:, 1]), len(xy), eps))


def hull(xy_sorted):
    x, y = _c(xy_sorted[:, 0]), _c(xy_sorted[:, 1])
    idx = np.zeros(max(len(x), 1), np.int64)
    h = lib().orc_hull(x, y, len(x), idx)
    return idx[:h]


def polar(xy, cx, cy):
    x, y = _c(xy[:, 0]), _c(xy[:, 1])
    th, r = np.zeros_like(x), np.zeros_like(x)
    lib().orc_polar(x, y, len(x), cx, cy, th, r)
    return th, r


def angle_max(xy, cx, cy, bins=720):
    sel = np.zeros(bins, np.int64)
    lib().orc_angle_max(_c(xy[:, 0]), _c(xy[:, 1]), len(xy), cx, cy, bins, sel)
    return sel


def residuals(xy, cx, cy, rad, bins=0, lo=0.0, hi=1.0, per_point=True):
    x, y = _c(xy[:, 0]), _c(xy[:, 1])
    n = len(x)
    th = np.zeros(n) if per_point else None
    dr = np.zeros(n) if per_point else None
    edges = np.linspace(lo, hi, bins + 1) if bins > 0 else np.zeros(1)
    hist = np.zeros(max(bins, 1), np.int64)
    rmm, amm = np.zeros(2), np.zeros(2, np.int64)
    lib().orc_residuals(x, y, n, cx, cy, rad, th.ctypes.data if per_point else None,
                        dr.ctypes.data if per_point else None, bins, edges, hist, rmm, amm)
    return dict(theta=th, dr=dr, hist=hist[:bins], rmax=rmm[0], rmin=rmm[1], argmax=int(amm[0]), argmin=int(amm[1]))


def ovality(xy, cx, cy, rad):
    x, y = _c(xy[:, 0]), _c(xy[:, 1])
    K = max(1024, len(x))
    scal, dev = np.zeros(5), np.zeros(K)
    lib().orc_ovality(x, y, len(x), cx, cy, rad, scal, dev.ctypes.data)
    return scal, dev


def fit_residual_batch(x, y, offsets, bins, lo, hi):
    x, y, offsets = _c(x), _c(y), _c(offsets, np.int64)
    S = len(offsets) - 1
    edges = np.linspace(lo, hi, bins + 1)
    fit = np.zeros((S, 3))
    hist = np.zeros((S, bins), np.int64)
    rmm, amm = np.zeros((S, 2)), np.zeros((S, 2), np.int64)
    lib().orc_fit_residual_batch(x, y, offsets, S, bins, edges, fit, hist, rmm, amm)
    return fit, hist, rmm, amm
