"""What the compiler made of the kernels (no GPU needed): the build keeps hipcc's per-kernel resource remarks in
pointcloud_amd/csrc/build/*.resources."""
import os
import re

import pytest

BUILD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pointcloud_amd", "csrc", "build")


def kernels(path):
    """{kernel name: {"VGPRs": n, "ScratchSize [bytes/lane]": n, ...}} from a resource-remark file."""
    out, name = {}, None
    for line in open(path):
        m = re.search(r"remark:\s+(?:Function Name|Name): (\S+)", line)
        if m:
            name = m.group(1)
            out[name] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z][^:]*): (\d+)", line)
        if m and name:
            out[name][m.group(1).strip()] = int(m.group(2))
    return out


def test_fit_kernels_keep_their_working_set_in_registers():
    """The fit kernels hold a slice in registers and LDS, two workgroups to a CU: no scratch memory, and registers
    for two waves per SIMD.  (A change that pushed the one-launch kernel to 256 registers and 120 bytes of scratch
    per lane cost a third of its speed without failing any parity test.)"""
    path = os.path.join(BUILD, "fit.resources")
    if not os.path.exists(path):
        pytest.skip("no resource remarks: build with make -C pointcloud_amd/csrc")
    ks = {k: v for k, v in kernels(path).items() if "kernel" in k}
    assert any("fit_residual_kernelILi16ELi16ELi256E" in k for k in ks)
    for name, res in ks.items():
        assert res.get("VGPRs Spill", 0) == 0, name
        assert res.get("VGPRs", 0) <= 256, name
        if "ILi16ELi16ELi256E" in name:             # (elsewhere a few bytes of stack serve a called function)
            assert res.get("ScratchSize [bytes/lane]", 0) == 0, name
