"""Per basic block of one kernel in a device assembly listing: vector registers touched, vector instructions,
spill stores / reloads (hipcc -S --offload-device-only ... -o fit.s; python tools/asm_blocks.py fit.s <mangled name part>)."""
import re
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2]
m = re.search(r"^(\S*%s\S*):" % re.escape(pat), txt, re.M)
i = m.start()
j = txt.index(".Lfunc_end", i)
body = txt[i:j].splitlines()
print(m.group(1), len(body), "lines")


def regs(l):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", l):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", l):
        out.add(int(m.group(1)))
    return out


blocks = []
cur = ["entry", 0, set(), 0, 0, 0]
for k, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur)
        cur = [m.group(1), k, set(), 0, 0, 0]
        continue
    t = l.strip()
    cur[2] |= regs(t)
    if t.startswith("v_"):
        cur[3] += 1
    if t.startswith("scratch_store"):
        cur[4] += 1
    if t.startswith("scratch_load"):
        cur[5] += 1
blocks.append(cur)
thr = int(sys.argv[3]) if len(sys.argv) > 3 else 60
for b in blocks:
    if len(b[2]) > thr or b[4] or b[5]:
        print(f"{b[0]:14s} line {b[1]:5d}  vgprs {len(b[2]):3d} (max v{max(b[2]) if b[2] else -1})  valu {b[3]:4d}  spill st {b[4]:3d} ld {b[5]:3d}")
