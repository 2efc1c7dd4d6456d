"""ctypes binding of libpcfit_hip.so (the C ABI declared in include/pcfit.h).

There is no CPU fallback: if the shared library is missing, or a call fails,
an exception is raised.  Host-pointer entry points take numpy arrays; the
``*_dev`` entry points are exposed raw (``lib().pcfit_*_dev``) for callers that
manage device memory themselves (bench.py does, through torch).
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(HERE, "libpcfit_hip.so")


class PcfitError(RuntimeError):
    pass


_lib = None

c_f64p = C.POINTER(C.c_double)
_vp = C.c_void_p
_i64 = C.c_int64
_i32 = C.c_int32
_f64 = C.c_double
_sz = C.c_size_t
_u64 = C.c_uint64

# name -> (restype, argtypes); every symbol include/pcfit.h declares
SIGNATURES = {
    "pcfit_version": (C.c_int, []),
    "pcfit_last_error": (C.c_char_p, []),
    "pcfit_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "pcfit_set_device": (C.c_int, [C.c_int]),
    "pcfit_get_device": (C.c_int, [C.POINTER(C.c_int)]),
    "pcfit_malloc": (C.c_int, [C.POINTER(_vp), _sz]),
    "pcfit_free": (C.c_int, [_vp]),
    "pcfit_memcpy_h2d": (C.c_int, [_vp, _vp, _sz, _vp]),
    "pcfit_memcpy_d2h": (C.c_int, [_vp, _vp, _sz, _vp]),
    "pcfit_memset": (C.c_int, [_vp, C.c_int, _sz, _vp]),
    "pcfit_stream_synchronize": (C.c_int, [_vp]),
    "pcfit_release_cache": (C.c_int, []),
    "pcfit_seed_state": (C.c_int, [_u64, _vp]),
    "pcfit_choice3_table": (C.c_int, [_vp, _i64, _i32, _vp]),
    "pcfit_bin_workspace_bytes": (_sz, [_i64, _i64]),
    "pcfit_bin_count_dev": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _sz, _vp]),
    "pcfit_bin_scatter_dev": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pcfit_bin2_workspace_bytes": (_sz, [_i64, _i64, C.c_int, _i64]),
    "pcfit_bin2_count_dev": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _vp, C.c_int, _i64, _vp, _vp, _vp, _sz, _vp]),
    "pcfit_bin2_scatter_dev": (C.c_int, [_i64, _i64, _vp, C.c_int, _i64, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pcfit_kasa_batch_dev": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    "pcfit_lm_batch_dev": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _vp, _f64, _i32, _f64, _f64, _f64, _vp, _vp, _vp]),
    "pcfit_fit_batch_dev": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _f64, _i32, _f64, _f64, _f64, _vp, _vp, _vp, _vp,
                                      _vp]),
    "pcfit_fit_residuals_batch_dev": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _f64, _i32, _f64, _f64, _f64, _vp, _vp, _vp,
                                                _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "pcfit_split_workspace_bytes": (_sz, [_i64, _i64, _i32]),
    "pcfit_kasa_split_dev": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _sz, _vp]),
    "pcfit_lm_split_dev": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _vp, _f64, _i32, _f64, _f64, _f64, _vp, _vp, _vp, _sz,
                                     _vp]),
    "pcfit_fit_split_dev": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _f64, _i32, _f64, _f64, _f64, _vp, _vp, _vp, _vp, _vp,
                                      _sz, _vp]),
    "pcfit_residuals_split_dev": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp,
                                            _vp, _sz, _vp]),
    "pcfit_angle_max_split_dev": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _vp, _i32, _vp, _vp, _sz, _vp]),
    "pcfit_ransac_workspace_bytes": (_sz, [_i64]),
    "pcfit_ransac_batch_dev": (C.c_int, [_vp, _vp, _vp, _i64, _f64, _i32, _f64, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                         _vp, _vp, _sz, _vp]),
    "pcfit_ransac_batch_static_dev": (C.c_int, [_vp, _vp, _vp, _i64, _f64, _i32, _f64, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                                _vp, _vp]),
    "pcfit_residuals_batch_dev": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "pcfit_angle_max_batch_dev": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _i32, _vp, _vp]),
    "pcfit_boundary_residuals_batch_dev": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "pcfit_boundary_residuals_batch": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp]),
    "pcfit_unique_hull_workspace_bytes": (_sz, [_i64, _i64]),
    "pcfit_unique_hull_batch_dev": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz,
                                              _vp]),
    "pcfit_ovality_batch_dev": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _i64, _vp]),
    "pcfit_ovality_workspace_bytes": (_sz, [_i64, _i64]),
    "pcfit_ovality_long_dev": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _vp, _sz, _vp]),
    "pcfit_kasa_batch": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "pcfit_lm_batch": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _f64, _i32, _f64, _f64, _f64, _vp, _vp]),
    "pcfit_ransac_batch": (C.c_int, [_vp, _vp, _vp, _i64, _f64, _i32, _f64, _vp, _u64, _vp, _vp, _vp, _vp, _vp]),
    "pcfit_residuals_batch": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "pcfit_angle_max_batch": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _i32, _vp]),
    "pcfit_bin_slices": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp]),
    "pcfit_unique_hull_batch": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pcfit_ovality_batch": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64]),
    "pcfit_qhull_order_batch": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp]),
    "pcfit_qhull_start_batch_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _sz, _vp, _vp, _vp, _vp,
                                              _i64, _vp]),
    "pcfit_debug_qhull_reasons": (C.c_int, [_vp, C.c_int]),
    "pcfit_rank_check_batch": (C.c_int, [_vp, _vp, _vp, _i64, _f64, _vp]),
    "pcfit_circle3_batch": (C.c_int, [_vp, _i64, _vp, _vp]),
    "pcfit_hybrid_batch": (C.c_int, [_vp, _vp, _vp, _i64, _f64, _i32, _f64, _vp, _f64, _i32, _vp, _i32, _u64] + [_vp] * 14
                           + [_i64]),
    "pcfit_count_text_lines": (C.c_int, [_vp, _sz, _i32, _vp]),
    "pcfit_format_csv_rows": (C.c_int, [_vp, _i64, _i32, _vp, _i32, _vp, _sz, _vp]),
    "pcfit_parse_xyz_text": (C.c_int, [_vp, _sz, _i32, _vp, _vp, _i64, _vp]),
    "pcfit_scan_upload": (C.c_int, [_vp, _i64, C.POINTER(_vp), _vp]),
    "pcfit_scan_free": (C.c_int, [_vp]),
    "pcfit_host_z_range": (C.c_int, [_vp, _i64, _i32, _vp]),
    "pcfit_host_window_counts": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _i32, _vp]),
    "pcfit_scan_upload_z_range": (C.c_int, [_vp, _i64, _f64, _f64, C.POINTER(_vp), _vp]),
    "pcfit_scan_bin_slices": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp]),
    "pcfit_scan_window_counts": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "pcfit_scan_hybrid_resident": (C.c_int, [_vp, _vp, _vp, _i64, _f64, _i32, _f64, _vp, _f64, _i32, _i32, _u64]
                                   + [_vp] * 13 + [_i64, _i64]),
    "pcfit_scan_hybrid_batch": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _f64, _i32, _f64, _vp, _f64, _i32, _i32, _u64]
                                + [_vp] * 13 + [_i64, _i64]),
}


def lib():
    """Load libpcfit_hip.so once; raise if it is not there (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise PcfitError(
                f"{SO_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950); this package has no CPU fallback")
        # Where PyTorch is installed its HIP runtime has to be the first one in the process: loaded after this
        # library's, torch finds no GPU ("No HIP GPUs are available"); loaded before, both share it.  So torch is
        # imported here first when it is there and has not been yet (PCFIT_TORCH_FIRST=0 leaves that out).
        if "torch" not in sys.modules and os.environ.get("PCFIT_TORCH_FIRST", "1") != "0":
            import importlib.util
            if importlib.util.find_spec("torch") is not None:
                import torch  # noqa: F401
        L = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int):
    if rc != 0:
        raise PcfitError(f"pcfit error {rc}: {lib().pcfit_last_error().decode(errors='replace')}")


def current_device() -> int:
    """The HIP device of the calling thread (a new thread starts on device 0)."""
    d = C.c_int(0)
    check(lib().pcfit_get_device(C.byref(d)))
    return d.value


def set_device(device: int):
    check(lib().pcfit_set_device(int(device)))


def _arr(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _p(a):
    return None if a is None else a.ctypes.data


def as_csr(x, y, offsets):
    x = _arr(x, np.float64).ravel()
    y = _arr(y, np.float64).ravel()
    if offsets is None:
        offsets = np.array([0, x.size], dtype=np.int64)
    offsets = _arr(offsets, np.int64)
    if x.size != y.size or offsets[-1] != x.size or offsets[0] != 0:
        raise ValueError("x, y and offsets do not describe the same points")
    return x, y, offsets


# ---------------------------------------------------------------- host calls
def seed_state(seed: int) -> np.ndarray:
    out = np.zeros(4, np.uint64)
    check(lib().pcfit_seed_state(int(seed) & (2**64 - 1), _p(out)))
    return out


def seed_state_from_numpy(seed) -> np.ndarray:
    """PCG64 state for any seed numpy accepts (numpy does the seeding)."""
    st = np.random.PCG64(seed).state["state"]
    m = 2**64 - 1
    return np.array([st["state"] >> 64, st["state"] & m, st["inc"] >> 64, st["inc"] & m], dtype=np.uint64)


def choice3_table(n: int, trials: int, seed=42) -> np.ndarray:
    out = np.zeros((trials, 3), np.int32)
    if isinstance(seed, (int, np.integer)) and 0 <= int(seed) < 2**64:
        state = seed_state(seed)
    else:
        state = seed_state_from_numpy(seed)
    check(lib().pcfit_choice3_table(_p(state), n, trials, _p(out)))
    return out


def kasa_batch(x, y, offsets=None):
    x, y, offsets = as_csr(x, y, offsets)
    S = offsets.size - 1
    out = np.full((S, 3), np.nan)
    status = np.zeros(S, np.int32)
    check(lib().pcfit_kasa_batch(_p(x), _p(y), _p(offsets), S, _p(out), _p(status)))
    return out, status


def lm_batch(x, y, offsets, init, huber_delta=1.0, max_iter=100, lm_lambda0=1e-2, tol_update=1e-10, tol_cost=1e-12):
    x, y, offsets = as_csr(x, y, offsets)
    S = offsets.size - 1
    init = _arr(init, np.float64).reshape(S, 3)
    out = np.zeros((S, 3))
    iters = np.zeros((S, 4), np.int32)
    check(lib().pcfit_lm_batch(_p(x), _p(y), _p(offsets), S, _p(init), huber_delta, max_iter, lm_lambda0, tol_update,
                               tol_cost, _p(out), _p(iters)))
    return out, iters


def ransac_batch(x, y, offsets=None, inlier_tol=0.5, max_trials=4000, min_inliers_frac=0.1, r_range=None, seed=42):
    x, y, offsets = as_csr(x, y, offsets)
    S = offsets.size - 1
    model = np.zeros((S, 3))
    score = np.zeros(S, np.int32)
    best = np.zeros(S, np.int32)
    mask = np.zeros(max(x.size, 1), np.uint8)
    status = np.zeros(S, np.int32)
    rr = None if r_range is None else _arr(r_range, np.float64)
    check(lib().pcfit_ransac_batch(_p(x), _p(y), _p(offsets), S, inlier_tol, max_trials, min_inliers_frac, _p(rr),
                                   int(seed), _p(model), _p(score), _p(best), _p(mask), _p(status)))
    return dict(model=model, score=score, best_trial=best, mask=mask[:x.size].astype(bool), status=status)


def residuals_batch(x, y, offsets, cxcyr, bins=0, lo=0.0, hi=1.0, want_theta=True, want_dr=True, want_radius=False):
    x, y, offsets = as_csr(x, y, offsets)
    S = offsets.size - 1
    fit = _arr(cxcyr, np.float64).reshape(S, 3)
    theta = np.zeros(x.size) if want_theta else None
    dr = np.zeros(x.size) if want_dr else None
    rad = np.zeros(x.size) if want_radius else None
    edges = np.linspace(lo, hi, bins + 1) if bins > 0 else None      # numpy.histogram's own edges
    hist = np.zeros((S, bins), np.int64) if bins > 0 else None
    rmm = np.zeros((S, 2))
    amm = np.zeros((S, 2), np.int64)
    check(lib().pcfit_residuals_batch(_p(x), _p(y), _p(offsets), S, _p(fit), _p(theta), _p(dr), _p(rad), bins,
                                      _p(edges), _p(hist), _p(rmm), _p(amm)))
    return dict(theta=theta, dr=dr, radius=rad, hist=hist, rmax=rmm[:, 0], rmin=rmm[:, 1], argmax=amm[:, 0], argmin=amm[:, 1])


def angle_max_batch(x, y, offsets, cxcyr, bins=720):
    x, y, offsets = as_csr(x, y, offsets)
    S = offsets.size - 1
    fit = _arr(cxcyr, np.float64).reshape(S, 3)
    sel = np.zeros((S, bins), np.int64)
    check(lib().pcfit_angle_max_batch(_p(x), _p(y), _p(offsets), S, _p(fit), bins, _p(sel)))
    return sel


def boundary_residuals_batch(x, y, offsets, cxcyr, bins=720, hist_bins=0, lo=0.0, hi=1.0):
    """angle_max_batch and the statistics of residuals_batch (no per-point outputs) about the same circles, each
    slice read once (pcfit_boundary_residuals_batch)."""
    x, y, offsets = as_csr(x, y, offsets)
    S = offsets.size - 1
    fit = _arr(cxcyr, np.float64).reshape(S, 3)
    sel = np.zeros((S, bins), np.int64)
    edges = np.linspace(lo, hi, hist_bins + 1) if hist_bins > 0 else None
    hist = np.zeros((S, hist_bins), np.int64) if hist_bins > 0 else None
    rmm = np.zeros((S, 2))
    amm = np.zeros((S, 2), np.int64)
    check(lib().pcfit_boundary_residuals_batch(_p(x), _p(y), _p(offsets), S, _p(fit), bins, _p(sel), hist_bins,
                                               _p(edges), _p(hist), _p(rmm), _p(amm)))
    return dict(sel=sel, hist=hist, rmax=rmm[:, 0], rmin=rmm[:, 1], argmax=amm[:, 0], argmin=amm[:, 1])


def bin_slices(points, z_lo, z_hi, want_perm=True):
    """Slices of ``points`` ((N,3) array, or a DeviceScan already on the GPU) in the windows [z_lo, z_hi]:
    (offsets, x, y, perm or None)."""
    resident = isinstance(points, DeviceScan)
    if resident:
        if not points._handle:
            raise PcfitError("the DeviceScan was closed")
        N = points.n_points
    else:
        pts = _arr(points, np.float64)
        if pts.ndim != 2 or pts.shape[1] != 3:
            raise ValueError("points must be (N,3)")
        N = pts.shape[0]
    z_lo = _arr(z_lo, np.float64)
    z_hi = _arr(z_hi, np.float64)
    S = z_lo.size
    cap = 2 * N + 1
    offsets = np.zeros(S + 1, np.int64)
    xo, yo = np.empty(cap), np.empty(cap)
    perm = np.empty(cap, np.int64) if want_perm else None
    total = C.c_int64(0)
    if resident:
        check(lib().pcfit_scan_bin_slices(points._handle, _p(z_lo), _p(z_hi), S, _p(offsets), _p(xo), _p(yo), _p(perm),
                                          cap, C.addressof(total)))
    else:
        check(lib().pcfit_bin_slices(_p(pts), N, _p(z_lo), _p(z_hi), S, _p(offsets), _p(xo), _p(yo), _p(perm),
                                     cap, C.addressof(total)))
    t = total.value
    return offsets, xo[:t], yo[:t], (perm[:t] if want_perm else None)


HULL_START_QHULL = -2      # PCFIT_HULL_START_QHULL


def _hull_starts(hull_start, S):
    """None, per-slice indices, or "qhull": the vertex scipy's ConvexHull lists first, for every slice."""
    if hull_start is None:
        return None
    if isinstance(hull_start, str):
        if hull_start != "qhull":
            raise ValueError('hull_start: indices, None or "qhull"')
        return np.full(S, HULL_START_QHULL, np.int64)
    return _arr(hull_start, np.int64)


def qhull_order_batch(ux, uy, offsets=None, n_unique=None, n_threads=0, want_walk=True):
    """The order of ``scipy.spatial.ConvexHull(xy).vertices`` for slices of distinct points (host cores):
    start, sure, and with want_walk the whole vertex lists (walk, n_walk)."""
    ux, uy, offsets = as_csr(ux, uy, offsets)
    S = offsets.size - 1
    nu = None if n_unique is None else _arr(n_unique, np.int64)
    start = np.full(S, -1, np.int64)
    sure = np.zeros(S, np.uint8)
    walk = np.zeros(max(ux.size, 1), np.int64) if want_walk else None
    n_walk = np.zeros(S, np.int64)
    check(lib().pcfit_qhull_order_batch(_p(ux), _p(uy), _p(offsets), _p(nu), S, int(n_threads), _p(start), _p(sure),
                                        _p(walk), _p(n_walk)))
    return dict(start=start, sure=sure.astype(bool), walk=walk, n_walk=n_walk, offsets=offsets)


HULL_ORDERS = {"canonical": 0, "qhull": 1}      # PCFIT_HULL_ORDER_CANONICAL, PCFIT_HULL_ORDER_QHULL


def _hull_order(order) -> int:
    """The hull_order argument of the C ABI for "qhull" (the order of scipy's ConvexHull(xy).vertices, which is
    the reference's: the default of every Python layer) or "canonical" (from the lexicographic minimum)."""
    try:
        return HULL_ORDERS[order]
    except (KeyError, TypeError):
        raise ValueError('order: "qhull" or "canonical"') from None


def unique_hull_batch(x, y, offsets=None, hull_start=None):
    x, y, offsets = as_csr(x, y, offsets)
    S = offsets.size - 1
    ux, uy = np.zeros(max(x.size, 1)), np.zeros(max(x.size, 1))
    nu = np.zeros(S, np.int64)
    hidx = np.zeros(max(x.size, 1), np.int64)
    nh = np.zeros(S, np.int64)
    status = np.zeros(S, np.int32)
    hs = _hull_starts(hull_start, S)
    check(lib().pcfit_unique_hull_batch(_p(x), _p(y), _p(offsets), S, _p(ux), _p(uy), _p(nu), _p(hidx), _p(nh),
                                        _p(status), _p(hs)))
    return dict(ux=ux, uy=uy, n_unique=nu, hull_idx=hidx, n_hull=nh, status=status, offsets=offsets)


def ovality_batch(x, y, offsets, cxcyr, want_profile=True):
    x, y, offsets = as_csr(x, y, offsets)
    S = offsets.size - 1
    fit = _arr(cxcyr, np.float64).reshape(S, 3)
    K = int(max(1024, np.diff(offsets).max() if S else 0))
    scal = np.zeros((S, 5))
    dev = np.zeros((S, K)) if want_profile else None
    check(lib().pcfit_ovality_batch(_p(x), _p(y), _p(offsets), S, _p(fit), _p(scal), _p(dev), K))
    return scal, dev


def rank_check_batch(x, y, offsets=None, eps=1e-12):
    x, y, offsets = as_csr(x, y, offsets)
    S = offsets.size - 1
    ok = np.zeros(S, np.uint8)
    check(lib().pcfit_rank_check_batch(_p(x), _p(y), _p(offsets), S, eps, _p(ok)))
    return ok.astype(bool)


def circle3_batch(triples):
    tri = _arr(triples, np.float64).reshape(-1, 6)
    out = np.zeros((tri.shape[0], 3))
    ok = np.zeros(tri.shape[0], np.uint8)
    check(lib().pcfit_circle3_batch(_p(tri), tri.shape[0], _p(out), _p(ok)))
    return out, ok.astype(bool)


OV_CAP = 8192          # candidates per slice that the ovality kernel holds in LDS (pipeline.hip); more go through a workspace


def hybrid_batch(x, y, offsets=None, inlier_tol=0.5, max_trials=4000, min_inliers_frac=0.1, r_range=None,
                 huber_delta=1.0, use_hull=True, hull_start=None, seed=42, want_profile=True, want_unique=True,
                 order="qhull"):
    """fit_circle_hybrid_at_z + the metrics of process_slice for every slice of a CSR batch.  ``order``: the
    order of every slice's hull candidates, "qhull" (the reference's) or "canonical"; ``hull_start`` (per-slice
    indices, -1 = lexicographic minimum, HULL_START_QHULL; or "qhull") decides slice by slice instead."""
    x, y, offsets = as_csr(x, y, offsets)
    S = offsets.size - 1
    tot = max(int(x.size), 1)
    status = np.zeros(S, np.int32)
    fit = np.full((S, 3), np.nan)
    ux, uy = (np.zeros(tot), np.zeros(tot)) if want_unique else (None, None)
    n_unique = np.zeros(S, np.int64)
    cand_off = np.zeros(S + 1, np.int64)
    cx_, cy_ = np.zeros(tot), np.zeros(tot)
    mask = np.zeros(tot, np.uint8)
    radius, theta = np.zeros(tot), np.zeros(tot)
    amm = np.zeros((S, 2), np.int64)
    ov = np.zeros((S, 5))
    # rows of the resampled profile: as long as the longest candidate list can get (hull candidates beyond
    # OV_CAP vertices are refused by the library when a profile is wanted)
    longest = int(np.diff(offsets).max()) if S else 0
    K = int(max(1024, min(OV_CAP, longest) if use_hull else longest))
    dev = np.full((S, K), np.nan) if want_profile else None
    rr = None if r_range is None else _arr(r_range, np.float64)
    hs = _hull_starts(hull_start, S)
    check(lib().pcfit_hybrid_batch(_p(x), _p(y), _p(offsets), S, inlier_tol, max_trials, min_inliers_frac, _p(rr),
                                   huber_delta, 1 if use_hull else 0, _p(hs), _hull_order(order), int(seed), _p(status),
                                   _p(fit), _p(ux),
                                   _p(uy), _p(n_unique), _p(cand_off), _p(cx_), _p(cy_), _p(mask), _p(radius),
                                   _p(theta), _p(amm), _p(ov), _p(dev), K))
    nc = int(cand_off[-1])
    return dict(status=status, fit=fit, ux=ux, uy=uy, n_unique=n_unique, offsets=offsets, cand_offsets=cand_off,
                cand_x=cx_[:nc], cand_y=cy_[:nc], mask=mask[:nc].astype(bool), radius=radius[:nc], theta=theta[:nc],
                argmax=amm[:, 0], argmin=amm[:, 1], ov=ov, dev_uniform=dev, grid_len=K)


class DeviceScan:
    """A scan uploaded once and kept on the GPU (pcfit_scan_upload): ``n_points``, and ``z_min`` / ``z_max``
    as numpy's ``points[:, 2].min()`` / ``.max()``.  Free it with ``close()`` (or let it go out of scope)."""

    def __init__(self, points, z_range=None):
        """``z_range`` = (lo, hi): only the rows with lo <= z <= hi go to the GPU (in their order);
        z_min / z_max are then not computed (NaN)."""
        pts = _arr(points, np.float64)
        if pts.ndim != 2 or pts.shape[1] != 3:
            raise ValueError("points must be (N,3)")
        self.device = current_device()         # the handle belongs to this device
        self._handle = _vp()
        if z_range is not None:
            kept = C.c_int64(0)
            check(lib().pcfit_scan_upload_z_range(_p(pts), int(pts.shape[0]), float(z_range[0]), float(z_range[1]),
                                                  C.byref(self._handle), C.addressof(kept)))
            self.n_points = int(kept.value)
            self.z_min = self.z_max = float("nan")
            return
        self.n_points = int(pts.shape[0])
        zr = np.full(2, np.nan)
        check(lib().pcfit_scan_upload(_p(pts), self.n_points, C.byref(self._handle), _p(zr)))
        self.z_min, self.z_max = float(zr[0]), float(zr[1])

    def close(self):
        if getattr(self, "_handle", None):
            lib().pcfit_scan_free(self._handle)
            self._handle = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __len__(self):
        return self.n_points

    def window_counts(self, z_lo, z_hi):
        """Points per closed window [z_lo[s], z_hi[s]] (pcfit_scan_window_counts)."""
        if not self._handle:
            raise PcfitError("the DeviceScan was closed")
        z_lo = _arr(z_lo, np.float64)
        z_hi = _arr(z_hi, np.float64)
        counts = np.zeros(z_lo.size, np.int64)
        if z_lo.size:
            check(lib().pcfit_scan_window_counts(self._handle, _p(z_lo), _p(z_hi), z_lo.size, _p(counts)))
        return counts


def host_z_range(points, n_threads: int = 0):
    """numpy's points[:, 2].min(), .max() by the library's host threads (both NaN if any z is NaN)."""
    pts = _arr(points, np.float64)
    zr = np.full(2, np.nan)
    check(lib().pcfit_host_z_range(_p(pts), int(pts.shape[0]), n_threads, _p(zr)))
    return float(zr[0]), float(zr[1])


def host_window_counts(points, z_lo, z_hi, n_threads: int = 0):
    """Points per closed window [z_lo[s], z_hi[s]], counted on the host (several threads)."""
    pts = _arr(points, np.float64)
    z_lo = _arr(z_lo, np.float64)
    z_hi = _arr(z_hi, np.float64)
    counts = np.zeros(z_lo.size, np.int64)
    if z_lo.size:
        check(lib().pcfit_host_window_counts(_p(pts), int(pts.shape[0]), _p(z_lo), _p(z_hi), z_lo.size, n_threads,
                                             _p(counts)))
    return counts


def scan_hybrid_batch(points, z_lo, z_hi, inlier_tol=0.5, max_trials=4000, min_inliers_frac=0.1, r_range=None,
                      huber_delta=1.0, use_hull=True, seed=42, want_profile=False, cand_capacity=None, order="qhull"):
    """bin_slices + hybrid_batch without the round trip of the binned points: the scan (an (N,3) array, uploaded
    for this call, or a DeviceScan already on the GPU) is binned on the device and only per-slice results and
    candidates come back.  ``order``: "qhull" (the reference's candidate order) or "canonical", an
    argument of this call alone.  With hull candidates the candidate arrays start small and the call is repeated once
    with the needed size if they were too small."""
    if isinstance(points, DeviceScan):
        if not points._handle:
            raise PcfitError("the DeviceScan was closed")
        N = points.n_points
        head = (lib().pcfit_scan_hybrid_resident, (points._handle,))
    else:
        pts = _arr(points, np.float64)
        if pts.ndim != 2 or pts.shape[1] != 3:
            raise ValueError("points must be (N,3)")
        N = pts.shape[0]
        head = (lib().pcfit_scan_hybrid_batch, (_p(pts), N))
    z_lo = _arr(z_lo, np.float64)
    z_hi = _arr(z_hi, np.float64)
    S = z_lo.size
    if cand_capacity is None:
        cand_capacity = 2 * N if not use_hull else min(2 * N, max(1 << 20, 1024 * S))
    rr = None if r_range is None else _arr(r_range, np.float64)
    order_code = _hull_order(order)
    K = OV_CAP
    while True:
        cap = max(int(cand_capacity), 1)
        offsets = np.zeros(S + 1, np.int64)
        status = np.zeros(S, np.int32)
        fit = np.full((S, 3), np.nan)
        n_unique = np.zeros(S, np.int64)
        cand_off = np.zeros(S + 1, np.int64)
        cx_, cy_ = np.empty(cap), np.empty(cap)
        mask = np.empty(cap, np.uint8)
        radius, theta = np.empty(cap), np.empty(cap)
        amm = np.zeros((S, 2), np.int64)
        ov = np.zeros((S, 5))
        dev = np.full((S, K), np.nan) if want_profile else None
        rc = head[0](*head[1], _p(z_lo), _p(z_hi), S, inlier_tol, max_trials, min_inliers_frac, _p(rr), huber_delta,
                     1 if use_hull else 0, order_code, int(seed), _p(offsets), _p(status), _p(fit), _p(n_unique),
                     _p(cand_off), _p(cx_), _p(cy_), _p(mask), _p(radius), _p(theta), _p(amm), _p(ov), _p(dev), K, cap)
        if rc != 0 and int(cand_off[-1]) > cap:           # candidate arrays too small: the count is known now
            cand_capacity = int(cand_off[-1])
            continue
        check(rc)
        break
    nc = int(cand_off[-1])
    return dict(status=status, fit=fit, n_unique=n_unique, offsets=offsets, cand_offsets=cand_off,
                cand_x=cx_[:nc], cand_y=cy_[:nc], mask=mask[:nc].astype(bool), radius=radius[:nc], theta=theta[:nc],
                argmax=amm[:, 0], argmin=amm[:, 1], ov=ov, dev_uniform=dev, grid_len=K)


def parse_xyz_text(data, n_threads: int = 0):
    """Rows of the XYZ text ``data`` (bytes or any buffer, e.g. a memory-mapped file; see pcfit_parse_xyz_text):
    (rows (n,3), deferred (n,) bool)."""
    raw = np.frombuffer(data, np.uint8)
    n = C.c_int64(0)
    check(lib().pcfit_count_text_lines(_p(raw), raw.size, n_threads, C.addressof(n)))
    cap = n.value
    rows = np.empty((cap, 3))
    flag = np.empty(cap, np.uint8)
    check(lib().pcfit_parse_xyz_text(_p(raw), raw.size, n_threads, _p(rows), _p(flag), cap, C.addressof(n)))
    return rows[:n.value], flag[:n.value].astype(bool)


def format_csv_rows(columns, is_int, n_threads: int = 0) -> bytes:
    """CSV body ('%.6f' floats, integers, NaN -> empty) of equally long columns (see pcfit_format_csv_rows)."""
    cols = np.ascontiguousarray(np.stack([np.asarray(c, np.float64) for c in columns]))
    flags = np.ascontiguousarray(is_int, np.uint8)
    n_cols, n_rows = cols.shape
    cap = max(n_rows * n_cols * 14, 64)
    n = _sz(0)
    while True:
        out = np.empty(cap, np.uint8)
        rc = lib().pcfit_format_csv_rows(_p(cols), n_rows, n_cols, _p(flags), n_threads, _p(out), cap, C.addressof(n))
        if rc != 0 and n.value > cap:
            cap = n.value
            continue
        check(rc)
        return out[:n.value].tobytes()
