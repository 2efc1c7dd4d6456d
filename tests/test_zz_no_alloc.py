"""pcfit_ransac_batch_static_dev: the RANSAC entry point that takes no workspace and allocates nothing
(its per-slice scratch is a fixed array inside the library's device image) against
pcfit_ransac_batch_dev, which takes a caller-allocated workspace.  Both run through the C ABI on the
same device buffers; every output must be identical (integer results and masks bit for bit, the model
as the same bits).  Two calls in a row, the second with a smaller batch, so that what the first left in
the shared scratch must not leak into the second; then the size limit."""
import ctypes as C

import numpy as np
import pytest

from pointcloud_amd import _ffi as F
from pointcloud_amd.synth import make_slices

pytestmark = pytest.mark.gpu


def _run_both(S, n, trials, seed):
    import torch
    dev = torch.device("cuda", 0)
    L = F.lib()
    x, y, off = make_slices(S, n, seed=seed, outlier_frac=0.3, jitter=0.05)
    # ragged batch: drop a few points of every third slice, and make one slice too short to fit
    lens = np.diff(off).copy()
    lens[::3] -= 7
    lens[S // 2] = 2
    keep = np.concatenate([np.arange(off[s], off[s] + lens[s]) for s in range(S)])
    x, y = x[keep], y[keep]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    sizes = sorted(set(int(v) for v in lens if v >= 3))
    table_of = np.array([sizes.index(int(v)) if v >= 3 else 0 for v in lens], np.int32)
    tables = np.concatenate([F.choice3_table(m, trials) for m in sizes]).astype(np.int32)

    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    p = lambda a: C.c_void_p(a.data_ptr())
    dx, dy, doff, dtab, dof = t(x), t(y), t(off), t(tables), t(table_of)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def outputs():
        return dict(model=torch.full((S, 3), -7.0, dtype=torch.float64, device=dev),
                    score=torch.full((S,), -7, dtype=torch.int32, device=dev),
                    best=torch.full((S,), -7, dtype=torch.int32, device=dev),
                    mask=torch.full((x.size,), 9, dtype=torch.uint8, device=dev),
                    status=torch.full((S,), -7, dtype=torch.int32, device=dev))

    a, b = outputs(), outputs()
    ws = torch.empty(L.pcfit_ransac_workspace_bytes(S), dtype=torch.uint8, device=dev)
    F.check(L.pcfit_ransac_batch_dev(p(dx), p(dy), p(doff), S, 0.5, trials, 0.1, None, p(dtab), p(dof), p(a["model"]),
                                     p(a["score"]), p(a["best"]), p(a["mask"]), p(a["status"]), p(ws), ws.numel(),
                                     stream))
    torch.cuda.synchronize()

    def static_call(o):
        F.check(L.pcfit_ransac_batch_static_dev(p(dx), p(dy), p(doff), S, 0.5, trials, 0.1, None, p(dtab), p(dof),
                                                p(o["model"]), p(o["score"]), p(o["best"]), p(o["mask"]), p(o["status"]),
                                                stream))
        torch.cuda.synchronize()

    # "allocates nothing" is read off the device itself: the free bytes hipMemGetInfo reports (torch's own counter
    # would not see a hipMalloc inside the library) are the same before and after a call - the second one, so
    # that the address of the library's array has been looked up already; its outputs were allocated before
    warm = outputs()
    static_call(warm)
    free_before = torch.cuda.mem_get_info()[0]
    static_call(b)
    assert torch.cuda.mem_get_info()[0] == free_before
    for k in b:
        assert torch.equal(warm[k], b[k]) or (b[k].dtype == torch.float64 and
                                              torch.equal(warm[k].view(torch.int64), b[k].view(torch.int64))), k
    for k in a:
        ha, hb = a[k].cpu().numpy(), b[k].cpu().numpy()
        if ha.dtype == np.float64:
            assert np.array_equal(ha.view(np.int64), hb.view(np.int64)), k      # same bits, NaN included
        else:
            assert np.array_equal(ha, hb), k
    sc = a["score"].cpu().numpy()
    assert sc[S // 2] == -1 and (np.delete(sc, S // 2) > 0).all()
    return sc


def test_static_scratch_ransac_equals_workspace_ransac_twice_second_smaller():
    first = _run_both(S=96, n=700, trials=300, seed=11)
    second = _run_both(S=17, n=260, trials=130, seed=12)          # fewer, shorter slices: stale keys must not show
    assert first.size == 96 and second.size == 17


def test_static_scratch_ransac_refuses_more_slices_than_its_array():
    import torch
    dev = torch.device("cuda", 0)
    L = F.lib()
    S = 65536 + 1
    one = torch.zeros(8, dtype=torch.float64, device=dev)
    p = lambda a: C.c_void_p(a.data_ptr())
    rc = L.pcfit_ransac_batch_static_dev(p(one), p(one), p(one), S, 0.5, 0, 0.1, None, None, None, p(one), p(one),
                                         p(one), p(one), p(one), None)
    assert rc == -4 and b"at most 65536" in L.pcfit_last_error()
