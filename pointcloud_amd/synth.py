"""Synthetic noisy cylinders for benchmarks and smoke checks (numpy, seeded)."""
from __future__ import annotations

import numpy as np


def make_slices(slices, per_slice, seed, radius=10.0, centre=(3.0, -2.0), sigma=0.05, outlier_frac=0.0, jitter=0.0):
    """Pre-binned slices in CSR form: x, y of slices*per_slice points and uniform
    offsets.  Points lie on a circle with gaussian radial noise; a fraction is
    replaced by radii uniform in [0, 1.4 R]; ``jitter`` moves every slice's
    centre and radius a little."""
    gen = np.random.default_rng(seed)
    total = slices * per_slice
    ang = gen.uniform(0, 2 * np.pi, total)
    r = radius + gen.normal(0, sigma, total)
    if outlier_frac > 0:
        bad = gen.random(total) < outlier_frac
        r[bad] = gen.uniform(0, 1.4 * radius, int(bad.sum()))
    sid = np.repeat(np.arange(slices), per_slice)
    shift = gen.normal(0, jitter, (slices, 3)) if jitter > 0 else np.zeros((slices, 3))
    r = r + shift[sid, 2]
    x = centre[0] + shift[sid, 0] + r * np.cos(ang)
    y = centre[1] + shift[sid, 1] + r * np.sin(ang)
    return x, y, np.arange(slices + 1, dtype=np.int64) * per_slice


def make_cylinder(n, slices, seed, radius=10.0, centre=(3.0, -2.0), sigma=0.05, outlier_frac=0.0, z_step=2.0):
    """(n, 3) scan of a cylinder spanning ``slices`` windows of height z_step."""
    gen = np.random.default_rng(seed)
    z = gen.uniform(0, slices * z_step, n)
    ang = gen.uniform(0, 2 * np.pi, n)
    r = radius + gen.normal(0, sigma, n)
    bad = gen.random(n) < outlier_frac
    r[bad] = gen.uniform(0, 1.4 * radius, int(bad.sum()))
    if n >= 2:
        z[0], z[1] = 0.0, slices * z_step
    return np.stack([centre[0] + r * np.cos(ang), centre[1] + r * np.sin(ang), z], axis=1)
