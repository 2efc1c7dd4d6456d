"""Three bin_slices calls with perm on a resident scan (for a kernel trace; experiments)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_amd import _ffi as F, DeviceScan
from pointcloud_amd.synth import make_cylinder
NP, NS = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100_000_000, 2000)
pts = make_cylinder(NP, NS, seed=0)
zc = np.arange(1.0, 2.0 * NS - 1.0 + 1e-9, 2.0)
with DeviceScan(pts) as sc:
    for rep in range(3):
        off, x, y, perm = F.bin_slices(sc, zc - 1, zc + 1, want_perm=True)
print(off[-1], perm[:4])
