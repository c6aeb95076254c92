#!/usr/bin/env python3
"""Instruction histogram of the loops of one kernel in a hipcc -save-temps .s file.
usage: tools/isa_hist.py file.s <mangled-name-substring> [--all]
Finds the kernel body, lists every backward branch (loop) with its instruction count and mnemonic classes."""
import collections
import re
import sys


def classify(m):
    if m.startswith("v_pk_"): return "v_pk"
    if m in ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32"): return "trans"
    if "dpp" in m: return "dpp"
    if m.startswith("v_mov") or m.startswith("v_accvgpr"): return "v_mov"
    if m.startswith("v_readlane") or m.startswith("v_readfirstlane") or m.startswith("v_writelane") or m.startswith("v_permlane"): return "lane"
    if m.startswith("v_mfma"): return "mfma"
    if m.startswith("v_cndmask"): return "cndmask"
    if m.startswith("v_cmp"): return "v_cmp"
    if m.startswith("v_"): return "valu"
    if m.startswith("ds_"): return m.split("_b")[0] if False else "ds:" + m
    if m.startswith("global_") or m.startswith("buffer_") or m.startswith("flat_") or m.startswith("scratch_"): return "vmem:" + m
    if m.startswith("s_waitcnt"): return "s_waitcnt"
    if m.startswith("s_nop"): return "s_nop"
    if m.startswith("s_barrier"): return "s_barrier"
    if m.startswith("s_load") or m.startswith("s_buffer"): return "smem"
    if m.startswith("s_"): return "salu"
    return m


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().split(":")[0].endswith(key.split()[-1]) or (l.startswith("_Z") and key in l.split(":")[0] and ":" in l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end + 1]
    labels, insts = {}, []
    for l in body:
        s = l.strip()
        if not s or s.startswith(";") or s.startswith("."):
            m = re.match(r"^(\.LBB[\w]+):", s)
            if m: labels[m.group(1)] = len(insts)
            continue
        m = re.match(r"^(\.LBB[\w]+):", s)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        insts.append(s.split(";")[0].strip())
    print(f"kernel body: {len(insts)} instructions")
    loops = []
    for i, ins in enumerate(insts):
        m = re.match(r"s_cbranch\w*\s+(\.LBB\w+)|s_branch\s+(\.LBB\w+)", ins)
        if m:
            t = labels.get(m.group(1) or m.group(2))
            if t is not None and t <= i: loops.append((t, i))
    for t, i in loops:
        h = collections.Counter(classify(x.split()[0]) for x in insts[t:i + 1])
        nops = sum(int(x.split()[1]) + 1 for x in insts[t:i + 1] if x.startswith("s_nop"))
        print(f"\nloop [{t}, {i}] = {i - t + 1} instructions (s_nop wait states {nops})")
        for k, v in sorted(h.items(), key=lambda kv: -kv[1]):
            print(f"  {k:28s} {v}")
    if "--dump" in sys.argv:
        t, i = max(loops, key=lambda p: p[1] - p[0]) if "--outer" in sys.argv else min(loops, key=lambda p: -(p[1] - p[0]))
        for x in insts[t:i + 1]: print(x)


main()
