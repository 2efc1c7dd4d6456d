"""Host-to-device copy rates on the box (pageable vs pinned, staged through pinned chunks) and where the
time of the scan-level call goes (experiments)."""
import os, sys, time, threading
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_amd import _ffi as F
from pointcloud_amd.synth import make_cylinder

dev = torch.device("cuda", 0)
n = 20_000_000
a = np.random.default_rng(0).normal(size=(n, 3))
d = torch.empty((n, 3), dtype=torch.float64, device=dev)
T = time.perf_counter
for rep in range(2):
    t = T(); d.copy_(torch.from_numpy(a)); torch.cuda.synchronize(); dt = T() - t
    print(f"pageable numpy -> device: {a.nbytes/dt/1e9:.1f} GB/s ({dt*1e3:.1f} ms)")
p = torch.empty((n, 3), dtype=torch.float64, pin_memory=True)
t = T(); p.numpy()[:] = a; dt = T() - t
print(f"host memcpy into pinned (1 thread): {a.nbytes/dt/1e9:.1f} GB/s")
for rep in range(2):
    t = T(); d.copy_(p, non_blocking=True); torch.cuda.synchronize(); dt = T() - t
    print(f"pinned -> device: {a.nbytes/dt/1e9:.1f} GB/s ({dt*1e3:.1f} ms)")
# multi-threaded staging: K threads copy chunks into pinned buffers, each followed by an async H2D
for K in (2, 4, 8):
    chunk = 8 << 20
    flat = a.reshape(-1).view(np.uint8)
    dflat = d.view(torch.uint8).reshape(-1)
    stage = [torch.empty(chunk, dtype=torch.uint8, pin_memory=True) for _ in range(2 * K)]
    streams = [torch.cuda.Stream() for _ in range(K)]
    nchunks = (flat.size + chunk - 1) // chunk
    def work(k):
        evs = [None, None]
        with torch.cuda.stream(streams[k]):
            for j, c in enumerate(range(k, nchunks, K)):
                b = stage[2 * k + (j & 1)]
                if evs[j & 1] is not None:
                    evs[j & 1].synchronize()
                lo, hi = c * chunk, min(flat.size, (c + 1) * chunk)
                b.numpy()[:hi - lo] = flat[lo:hi]
                dflat[lo:hi].copy_(b[:hi - lo], non_blocking=True)
                e = torch.cuda.Event(); e.record(); evs[j & 1] = e
    for rep in range(2):
        t = T()
        th = [threading.Thread(target=work, args=(k,)) for k in range(K)]
        [x.start() for x in th]; [x.join() for x in th]; torch.cuda.synchronize(); dt = T() - t
    print(f"staged through pinned chunks, {K} threads: {a.nbytes/dt/1e9:.1f} GB/s ({dt*1e3:.1f} ms)")
    assert torch.equal(d.cpu(), torch.from_numpy(a))

# the scan-level call on the C1 shape
pts = make_cylinder(1_000_000, 200, seed=0)
zc = np.arange(1.0, 399.0 + 1e-9, 2.0)
F.scan_hybrid_batch(pts, zc - 1, zc + 1)
t = T(); h = F.scan_hybrid_batch(pts, zc - 1, zc + 1); dt = T() - t
print(f"scan_hybrid_batch 1M pts / {len(zc)} windows: {dt*1e3:.2f} ms")
from pointcloud_amd import Config, CylinderAnalyzer
az = CylinderAnalyzer(Config()); az.analyze_with_progress(pts)
t = T(); az.analyze_with_progress(pts); dt = T() - t
print(f"analyze_with_progress: {dt*1e3:.2f} ms")
