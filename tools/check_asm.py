"""Scan the gfx950 assembly of the kernel files for a scalar select or branch on SCC whose latest SCC
writer is a carry-producing add/sub: hipcc 7.2 emitted exactly that for a uniform 64-bit `n > cap ? cap : n`
(the compare went to the vector unit, the select stayed scalar)."""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "pointcloud_amd", "csrc")
writers = re.compile(r'^\s*(s_cmp|s_bitcmp|s_add|s_sub|s_and|s_or|s_xor|s_lshl|s_lshr|s_ashr|s_min|s_max|s_abs|s_not|s_andn2|s_orn2|s_nand|s_nor|s_xnor|s_bfe|s_addc|s_subb|s_absdiff|s_wqm|s_bcnt|s_ff|s_flbit|s_quadmask)')
bad = 0
for f in sys.argv[1:] or ["api", "fit", "split", "ransac", "residual", "bin", "hull", "pipeline"]:
    out = f"/tmp/check_{f}.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off",
                    "-fno-fast-math", f"-I{root}/include", f"-I{src}", "--cuda-device-only", "-S",
                    os.path.join(src, f + ".hip"), "-o", out], check=True, stderr=subprocess.DEVNULL)
    last = None
    for i, l in enumerate(open(out)):
        t = l.strip()
        if t.endswith(":") and not t.startswith(";"):
            last = None
        m = writers.match(l)
        if m and not t.startswith("s_cselect"):
            last = t
        if (t.startswith("s_cselect") or t.startswith("s_cbranch_scc")) and last and last.startswith(("s_addc", "s_subb")):
            bad += 1
            print(f"{f}.s:{i}: {last}  ->  {t}")
print("suspicious:", bad)
sys.exit(1 if bad else 0)
