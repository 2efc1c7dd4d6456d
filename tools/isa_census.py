"""Instruction census of one kernel in a gfx950 assembly listing, per basic block and per class:
    hipcc -O3 ... --cuda-device-only -S fit.hip -o fit.s
    python tools/isa_census.py fit.s <part of the mangled kernel name> [min instructions per block to print]
Classes: f64 = FP64 arithmetic that the algorithm asks for (add, mul, fma, min / max, rsq, rcp, sqrt, floor, ...),
cmp = vector compares, sel = v_cndmask, mov = v_mov / v_accvgpr moves, xch = lane exchanges (DPP moves, permlane swaps,
readlane / writelane), int = other vector integer / bit work, lds = ds_*, mem = global / scratch / buffer, s = scalar.
A block that is the target of a backward branch is marked as a loop head."""
import re
import sys
from collections import Counter, OrderedDict

F64 = re.compile(r"^v_(add|mul|fma|min|max|rsq|rcp|sqrt|floor|ceil|trunc|rndne|fract|ldexp|frexp_mant|div_scale|div_fmas|div_fixup|cvt)_?\w*f64")


def classify(t):
    op = t.split()[0]
    if op.startswith("s_"):
        return "s"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "scratch_", "buffer_", "flat_")):
        return "mem"
    if not op.startswith("v_"):
        return "other"
    if "dpp" in t or "row_" in t or "quad_perm" in t or op.startswith(("v_permlane", "v_readlane", "v_writelane", "v_readfirstlane")):
        return "xch"
    if op.startswith("v_cmp"):
        return "cmp"
    if op.startswith("v_cndmask"):
        return "sel"
    if op.startswith(("v_mov", "v_accvgpr", "v_swap")):
        return "mov"
    if F64.match(op) or op.startswith(("v_cvt_f64", "v_cvt_i32_f64", "v_cvt_u32_f64")):
        return "f64"
    return "int"


def blocks_of(path, pat):
    txt = open(path).read()
    m = re.search(r"^(\S*%s\S*):" % re.escape(pat), txt, re.M)
    body = txt[m.start():txt.index(".Lfunc_end", m.start())].splitlines()
    blocks = OrderedDict()
    cur = "entry"
    blocks[cur] = dict(line=0, ops=Counter(), text=[], targets=[])
    for k, l in enumerate(body):
        mm = re.match(r"^(\.LBB\d+_\d+):", l)
        if mm:
            cur = mm.group(1)
            blocks[cur] = dict(line=k, ops=Counter(), text=[], targets=[])
            continue
        t = l.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        t = t.split(";")[0].strip()
        if not t:
            continue
        blocks[cur]["ops"][classify(t)] += 1
        blocks[cur]["text"].append(t)
        mb = re.match(r"s_(c?branch\w*)\s+(\.LBB\d+_\d+)", t)
        if mb:
            blocks[cur]["targets"].append(mb.group(2))
    return m.group(1), blocks


CLS = ["f64", "cmp", "sel", "mov", "xch", "int", "lds", "mem", "s"]

if __name__ == "__main__":
    name, blocks = blocks_of(sys.argv[1], sys.argv[2])
    thr = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    order = list(blocks)
    pos = {b: i for i, b in enumerate(order)}
    heads = set()
    for b, d in blocks.items():
        for t in d["targets"]:
            if t in pos and pos[t] <= pos[b]:
                heads.add(t)
    print(name)
    print(f"{'block':12s} {'line':>6s} " + " ".join(f"{c:>5s}" for c in CLS) + "  valu  notes")
    tot = Counter()
    for b, d in blocks.items():
        o = d["ops"]
        tot.update(o)
        valu = sum(o[c] for c in ("f64", "cmp", "sel", "mov", "xch", "int"))
        if sum(o.values()) >= thr or b in heads:
            rsq = sum(1 for t in d["text"] if t.startswith("v_rsq_f64"))
            notes = []
            if b in heads:
                notes.append("LOOP")
            if rsq:
                notes.append(f"rsq x{rsq}")
            if any(t.startswith("s_barrier") for t in d["text"]):
                notes.append("barrier")
            if any(t.startswith("s_setprio") for t in d["text"]):
                notes.append("setprio")
            back = [t for t in d["targets"] if t in pos and pos[t] <= pos[b]]
            if back:
                notes.append("back->" + ",".join(back))
            print(f"{b:12s} {d['line']:6d} " + " ".join(f"{o[c]:5d}" for c in CLS) + f" {valu:5d}  " + " ".join(notes))
    print("total        " + "       " + " ".join(f"{tot[c]:5d}" for c in CLS))
