"""Text loader timing on the box (experiments)."""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_amd.utils.data_io import load_txt_points
from pointcloud_amd import Config, CylinderAnalyzer
from pointcloud_amd.synth import make_cylinder
pts = make_cylinder(2_000_000, 200, seed=0)
d = tempfile.mkdtemp(); p = os.path.join(d, "scan.txt")
np.savetxt(p, pts, fmt="%.6f")
print(f"file: {os.path.getsize(p)/1e6:.0f} MB, {len(pts)} lines, cores {os.cpu_count()}")
for th in (1, 4, 16, 0):
    t = time.perf_counter(); a = load_txt_points(p, th); dt = time.perf_counter() - t
    print(f"load_txt_points threads={th}: {dt*1e3:.1f} ms ({os.path.getsize(p)/dt/1e6:.0f} MB/s)")
os.chdir(d)
az = CylinderAnalyzer(Config())
az.analyze_file(p)
import io, contextlib
with contextlib.redirect_stdout(io.StringIO()):
    t = time.perf_counter(); res = az.analyze_file(p); dt = time.perf_counter() - t
print(f"analyze_file (load + analyse + two CSVs): {dt*1e3:.1f} ms, {len(res)} results")
