"""Timing of the fit / residual entry points (experiments; PHASES=1 uses the phase-clock build)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_amd import _ffi as F
from pointcloud_amd.synth import make_slices

S, n, H = int(os.environ.get("S", 10000)), int(os.environ.get("NP", 8192)), int(os.environ.get("H", 256))
dev = torch.device("cuda", 0)
xh, yh, offh = make_slices(S, n, seed=1, sigma=0.05, jitter=0.02)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
x, y, off, edges = t(xh), t(yh), t(offh), t(np.linspace(-0.5, 0.5, max(H, 1) + 1))
kasa = torch.empty(S, 3, dtype=torch.float64, device=dev); fit = torch.empty_like(kasa)
iters = torch.empty(S, 4, dtype=torch.int32, device=dev); status = torch.empty(S, dtype=torch.int32, device=dev)
hist = torch.empty(S, max(H, 1), dtype=torch.int64, device=dev); rmm = torch.empty(S, 2, dtype=torch.float64, device=dev)
amm = torch.empty(S, 2, dtype=torch.int64, device=dev)
p = lambda a: C.c_void_p(a.data_ptr())
if os.environ.get("PHASES"):
    F.SO_PATH = F.SO_PATH.replace("libpcfit_hip.so", "libpcfit_hip_phase.so")
if os.environ.get("SO"):                       # another build of the library, for A/B runs on one box
    F.SO_PATH = F.SO_PATH.replace("libpcfit_hip.so", os.environ["SO"])
L = F.lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
tag = f"S={S} n={n}"

def timeit(name, fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{tag:34s} {name:24s} {e0.elapsed_time(e1)/reps:8.3f} ms", flush=True)

if os.environ.get("ONLY") == "fused":
    for _ in range(3):
        F.check(L.pcfit_fit_residuals_batch_dev(p(x), p(y), p(off), S, n, 1.0, 100, 1e-2, 1e-10, 1e-12, p(kasa), p(fit), p(iters), p(status), H, p(edges), p(hist), p(rmm), p(amm), st))
    torch.cuda.synchronize()
    sys.exit(0)
timeit("fit (kasa+lm)", lambda: F.check(L.pcfit_fit_batch_dev(p(x), p(y), p(off), S, n, 1.0, 100, 1e-2, 1e-10, 1e-12, p(kasa), p(fit), p(iters), p(status), st)))
for mi in (100,):
    timeit(f"fit max_iter={mi}", lambda: F.check(L.pcfit_fit_batch_dev(p(x), p(y), p(off), S, n, 1.0, mi, 1e-2, 1e-10, 1e-12, p(kasa), p(fit), p(iters), p(status), st)))
    print("   sweeps", iters[0].tolist())
timeit("kasa only", lambda: F.check(L.pcfit_kasa_batch_dev(p(x), p(y), p(off), S, n, p(kasa), p(status), st)))
timeit("kasa+residuals fused", lambda: F.check(L.pcfit_fit_residuals_batch_dev(p(x), p(y), p(off), S, n, 1.0, 0, 1e-2, 1e-10, 1e-12, p(kasa), p(fit), p(iters), p(status), H, p(edges), p(hist), p(rmm), p(amm), st)))
timeit("fit+residuals fused", lambda: F.check(L.pcfit_fit_residuals_batch_dev(p(x), p(y), p(off), S, n, 1.0, 100, 1e-2, 1e-10, 1e-12, p(kasa), p(fit), p(iters), p(status), H, p(edges), p(hist), p(rmm), p(amm), st)))
if os.environ.get("PHASES"):
    out = (C.c_int64 * 16)()
    L2 = C.CDLL(F.SO_PATH)
    L2.pcfit_debug_phase_clocks(out, 1)
    F.check(L.pcfit_fit_residuals_batch_dev(p(x), p(y), p(off), S, n, 1.0, int(os.environ.get("MI", 100)), 1e-2, 1e-10, 1e-12, p(kasa), p(fit), p(iters), p(status), H, p(edges), p(hist), p(rmm), p(amm), st))
    torch.cuda.synchronize()
    L2.pcfit_debug_phase_clocks(out, 0)
    names = ["load", "kasa", "sweep points (+solve)", "sweep reduce", "lm end", "residual"]
    tot = sum(out[:6])
    sampled = (S + 63) // 64                       # one workgroup in 64 reports; wall_clock64 ticks at 100 MHz
    for k, nm in enumerate(names):
        print(f"   phase {nm:22s} {out[k]/sampled/100:8.2f} us/slice  {100*out[k]/max(tot,1):5.1f} %")
    print(f"   sum {tot/sampled/100:8.2f} us per slice and workgroup")
print("iters", iters[:2].tolist())
