"""CPU oracle for the slice -> circle fit -> boundary -> residual path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``pointcloud_amd/`` imports this
package.  It exists so that ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` have an independent statement of what the
reference computes:

* ``oracle.ref_np``   - numpy restatement that mirrors the reference's numpy /
  LAPACK calls one for one (so it inherits numpy's summation order, LAPACK's
  2x2 solves and the host libm).  Pinned against outputs of the real reference
  recorded in ``tests/golden/*.npz`` (see ``tests/golden/gen_golden.py``).
* ``oracle.native``   - ctypes binding of ``pcfit_oracle.c``: a plain C
  restatement with every floating-point operation spelled out (no LAPACK, no
  contraction), which is what the HIP kernels are compared with bit for bit.
"""
