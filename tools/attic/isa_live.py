#!/usr/bin/env python3
"""VGPR liveness over the largest inner loop of a kernel in a hipcc -save-temps .s file (straight-line, cyclic):
prints the number of live VGPRs before every instruction and the peak.  usage: isa_live.py file.s <kernel-substr> [--inner|--outer]"""
import re
import sys


def regs(tok):
    out = []
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1) is not None:
            out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.append(int(m.group(3)))
    return out


def split_ops(ins):
    parts = ins.split(None, 1)
    if len(parts) < 2:
        return parts[0], []
    ops = [o.strip() for o in re.split(r",(?![^\[]*\])", parts[1])]
    return parts[0], ops


NO_DST = ("global_store", "buffer_store", "ds_write", "global_atomic_add_f32", "s_", "v_cmp", "v_cmpx", "flat_store", "scratch_store", "ds_add")
RMW = ("v_fmac", "v_mac", "v_permlane", "v_writelane", "v_cndmask_b32_dpp")


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    labels, insts = {}, []
    for l in lines[start:end + 1]:
        s = l.strip()
        m = re.match(r"^(\.LBB[\w]+):", s)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        insts.append(s.split(";")[0].strip())
    loops = []
    for i, ins in enumerate(insts):
        m = re.match(r"s_cbranch\w*\s+(\.LBB\w+)|s_branch\s+(\.LBB\w+)", ins)
        if m:
            t = labels.get(m.group(1) or m.group(2))
            if t is not None and t <= i:
                loops.append((t, i))
    loops.sort(key=lambda p: p[1] - p[0])
    big = [p for p in loops if p[1] - p[0] > 100]
    t, e = big[0] if "--outer" not in sys.argv else big[-1]
    body = insts[t:e + 1]
    defs, uses = [], []
    for ins in body:
        mn, ops = split_ops(ins)
        d, u = [], []
        if ops and not mn.startswith(NO_DST):
            d = regs(ops[0])
            for o in ops[1:]:
                u += regs(o)
            if mn.startswith(RMW) or "dpp" in mn and mn.startswith(("v_fmac", "v_mul_f32_dpp", "v_add_f32_dpp", "v_mov_b32_dpp")):
                u += d
            if mn.startswith("v_permlane"):
                d = d + regs(ops[1])
        else:
            for o in ops:
                u += regs(o)
        defs.append(set(d))
        uses.append(set(u))
    n = len(body)
    live = set()
    for _ in range(3):  # cyclic fixpoint
        for i in range(n - 1, -1, -1):
            live = (live - defs[i]) | uses[i]
    counts = [0] * n
    for i in range(n - 1, -1, -1):
        live = (live - defs[i]) | uses[i]
        counts[i] = len(live)
    peak = max(counts)
    print(f"loop [{t}, {e}] {n} instructions, peak live VGPRs {peak}")
    step = max(1, n // 60)
    for i in range(0, n, step):
        print(f"{i:5d} live {counts[i]:4d}  {body[i][:90]}")
    pi = counts.index(peak)
    print("peak at", pi, body[pi])


main()
