"""Phase clocks of unique_hull_kernel on config-4-like slices through the analyzer's own path (hull only, interior
filter first).  Needs the phase build (make -C pointcloud_amd/csrc phase) for the clocks; PHASES=0 times only."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_amd import _ffi as F
PH = os.environ.get("PHASES", "1") == "1"
if PH:
    F.SO_PATH = os.path.join(os.path.dirname(F.SO_PATH), "libpcfit_hip_phase.so")
from pointcloud_amd.synth import make_slices
L = F.lib()
if PH:
    L.pcfit_debug_hull_phases.argtypes = [C.c_void_p, C.c_int]
for S, n in ((int(sys.argv[1]), int(sys.argv[2])),) if len(sys.argv) > 2 else ((600, 50000), (57, 50000)):
    x, y, off = make_slices(S, n, seed=5, outlier_frac=float(os.environ.get("OUTL", 0.1)))
    run = lambda: F.hybrid_batch(x, y, off, max_trials=16, want_unique=False, want_profile=False)
    h = run()
    out = np.zeros(8, np.int64)
    if PH:
        L.pcfit_debug_hull_phases(out.ctypes.data, 1)
    t0 = time.perf_counter(); h = run(); dt = time.perf_counter() - t0
    if PH:
        L.pcfit_debug_hull_phases(out.ctypes.data, 1)
    names = ["load+filter", "sort", "unique", "rank", "chains"]
    print(f"{S} x {n}: call {dt*1e3:.1f} ms; kept median {int(np.median(h['n_unique']))} max {int(h['n_unique'].max())}; hull median {int(np.median(np.diff(h['cand_offsets'])))}; "
          + "  ".join(f"{k} {out[i]/S/100:.1f}us" for i, k in enumerate(names)), flush=True)
