#!/usr/bin/env python3
"""Issue-cycle budget of a kernel's loops from its gfx950 ISA (hipcc -S --cuda-device-only).

usage: tools/isa_budget.py file.s <kernel-name-substring> [--loop N] [--per D] [--dump] [--prices occ]

Finds the kernel, lists every loop (a backward branch and its target) with its instruction count, and for the selected
loop (default: the longest) prints instruction class x count (x 1/D: e.g. D = 16 states per chunk) x a cycle price ->
VALU pipe cycles, next to VGPR bank statistics of the VALU sources (bank = register index mod 4: an instruction whose
VGPR sources share a bank pays extra operand-read cycles, tools/microbench/microbench5.hip V8 / V9: 4.2 vs 2.05 cycles).

Prices (cycles of one SIMD per wave-instruction) are the measured ones of profiles/r05_microbench_mix.txt; they are a model,
the PMC counters (SQ_ACTIVE_INST_VALU x 4 / SQ_INSTS_VALU) are the check.
"""
import collections
import re
import sys

TRANS = ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32")


SREG = re.compile(r"\bs\d+\b|\bs\[\d+:\d+\]|\b0x[0-9a-f]+\b|\bvcc\b|\bexec\b")


def mnem(ins):
    m = ins.split()[0]
    for suf in ("_e32", "_e64", "_sdwa"):
        if m.endswith(suf): m = m[:-len(suf)]
    return m


def classify(ins):
    m = mnem(ins)
    if m.startswith("v_pk_"): return "pk (2 lane-ops)"
    if m in TRANS: return "transcendental"
    if "_dpp" in m or " row_" in ins or "quad_perm" in ins or "row_bcast" in ins or "wave_sh" in ins: return "dpp"
    if m.startswith("v_permlane"): return "permlane swap"
    if m.startswith("v_readlane") or m.startswith("v_readfirstlane") or m.startswith("v_writelane"): return "readlane"
    if m.startswith("v_mov") or m.startswith("v_accvgpr"): return "v_mov"
    if m.startswith("v_cvt") or m.startswith("v_lshl") or m.startswith("v_lshr") or m.startswith("v_and") or m.startswith("v_or") or m.startswith("v_perm") or m.startswith("v_bfe") or m.startswith("v_bfi"): return "convert / bit"
    if m.startswith("v_cndmask") or m.startswith("v_cmp"): return "select / compare"
    if m.startswith("v_mfma"): return "mfma"
    if m.startswith("v_"):
        ops = ins.split(None, 1)[1].split(",")[1:] if " " in ins else []
        if any(SREG.search(o) for o in ops): return "scalar VALU, SGPR / literal operand"
        return "scalar fp32 / int VALU"
    if m.startswith("ds_"): return "LDS " + m
    if m.startswith(("global_", "buffer_", "flat_", "scratch_")): return "VMEM " + m
    if m.startswith("s_waitcnt"): return "s_waitcnt"
    if m.startswith("s_nop"): return "s_nop"
    if m.startswith("s_barrier"): return "s_barrier"
    if m.startswith(("s_load", "s_buffer")): return "SMEM"
    if m.startswith("s_"): return "SALU"
    return m


# cycles of the SIMD's VALU pipe per wave-instruction (profiles/r05_microbench_mix.txt; r01_microbench_issue.txt)
# profiles/r05_microbench_mix.txt, columns occ2 / occ4 (waves per SIMD): cycles of the SIMD per wave-instruction
PRICES = {2: {"pk (2 lane-ops)": 5.3, "transcendental": 8.8, "dpp": 5.3, "permlane swap": 8.8, "readlane": 3.0, "v_mov": 3.0, "convert / bit": 3.0,
              "select / compare": 3.0, "scalar fp32 / int VALU": 3.05, "scalar VALU, SGPR / literal operand": 5.5},
          4: {"pk (2 lane-ops)": 4.55, "transcendental": 8.4, "dpp": 4.64, "permlane swap": 8.45, "readlane": 2.6, "v_mov": 2.6, "convert / bit": 2.6,
              "select / compare": 2.6, "scalar fp32 / int VALU": 2.6, "scalar VALU, SGPR / literal operand": 4.8}}
PRICE = PRICES[4]

VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def vgpr_sources(ins):
    """VGPR indices read by a VALU instruction (first operand = destination, except for the *mac / *fmac forms which also read it)."""
    parts = ins.split(None, 1)
    if len(parts) < 2: return []
    ops = [o.strip() for o in parts[1].split(",")]
    m = parts[0]
    srcs = ops[1:]
    if "fmac" in m or "_mac_" in m or m.startswith("v_pk_fmac"): srcs = ops[1:] + ops[:1]
    out = []
    for o in srcs:
        mm = VREG.search(o)
        if not mm: continue
        if mm.group(1) is not None: out.append([int(mm.group(1))])
        else: out.append(list(range(int(mm.group(2)), int(mm.group(3)) + 1)))
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    path, key = args[0], args[1]
    opt = {a.split("=")[0]: (a.split("=")[1] if "=" in a else True) for a in sys.argv[1:] if a.startswith("--")}
    per = float(opt.get("--per", 1))
    global PRICE
    PRICE = PRICES[int(opt.get("--occ", 4))]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    meta_end = next(i for i in range(end, len(lines)) if ".end_amdhsa_kernel" in lines[i])
    meta = {k: v for k, v in (re.findall(r"\.amdhsa_(next_free_vgpr|next_free_sgpr|accum_offset|group_segment_fixed_size|private_segment_fixed_size)\s+(\d+)", "\n".join(lines[end:meta_end])))}
    labels, insts = {}, []
    for l in lines[start:end + 1]:
        s = l.strip()
        m = re.match(r"^(\.LBB[\w]+):", s)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"): continue
        insts.append(s.split(";")[0].strip())
    print(f"kernel {lines[start].split(':')[0]}\n  {len(insts)} instructions; " + ", ".join(f"{k} {v}" for k, v in meta.items()))
    loops = []
    for i, ins in enumerate(insts):
        m = re.match(r"s_cbranch\w*\s+(\.LBB\w+)|s_branch\s+(\.LBB\w+)", ins)
        if m:
            t = labels.get(m.group(1) or m.group(2))
            if t is not None and t <= i: loops.append((t, i))
    loops.sort(key=lambda p: p[0] - p[1])
    for k, (t, i) in enumerate(loops):
        nv = sum(1 for x in insts[t:i + 1] if x.startswith("v_"))
        print(f"  loop {k}: instructions [{t}, {i}] = {i - t + 1}, VALU {nv}")
    if not loops: return
    t, i = loops[int(opt.get("--loop", 0))]
    body = insts[t:i + 1]
    h = collections.Counter(classify(x) for x in body)
    nops = sum(int(x.split()[1]) + 1 for x in body if x.startswith("s_nop"))
    if "--nest" in opt:   # --nest=K:M: loop K lies inside the selected loop and runs M times per iteration of it
        k, m = (int(v) for v in opt["--nest"].split(":"))
        t2, i2 = loops[k]
        assert t <= t2 and i2 <= i, "the nested loop must lie inside the selected one"
        inner = insts[t2:i2 + 1]
        hi = collections.Counter(classify(x) for x in inner)
        for key, v in hi.items(): h[key] += (m - 1) * v
        nops += (m - 1) * sum(int(x.split()[1]) + 1 for x in inner if x.startswith("s_nop"))
        print(f"\n(loop [{t2}, {i2}] = {len(inner)} instructions counted {m} times)")
    print(f"\nloop [{t}, {i}]: {len(body)} instructions, s_nop wait states {nops}; counts per 1/{per:g} of an iteration")
    print(f"| class | per iteration | per 1/{per:g} | price (cyc) | pipe cycles per 1/{per:g} |\n|---|---|---|---|---|")
    tot_v = tot_c = 0
    for k, v in sorted(h.items(), key=lambda kv: -kv[1]):
        pr = PRICE.get(k)
        if pr is not None:
            tot_v += v
            tot_c += v * pr
        print(f"| {k} | {v} | {v / per:.1f} | {pr if pr is not None else ''} | {v * pr / per:.0f} |" if pr is not None else f"| {k} | {v} | {v / per:.1f} | | |")
    print(f"| **VALU total** | {tot_v} | {tot_v / per:.1f} | {tot_c / max(tot_v, 1):.2f} avg | {tot_c / per:.0f} |")
    # VGPR bank statistics
    conf = collections.Counter()
    for x in body:
        if not x.startswith("v_") or x.startswith("v_mfma"): continue
        srcs = vgpr_sources(x)
        if not srcs: continue
        if x.startswith("v_pk_"):
            # lanes-op halves: lo dwords and hi dwords are read together; count the worst half
            worst = 0
            for half in (0, 1):
                banks = collections.Counter(r[min(half, len(r) - 1)] % 4 for r in srcs)
                worst = max(worst, max(banks.values()))
            conf[("pk", len(srcs), worst)] += 1
        else:
            banks = collections.Counter(r[0] % 4 for r in srcs)
            conf[("1", len(srcs), max(banks.values()))] += 1
    print("\nVGPR source banks (kind, VGPR sources, max sources on one bank): count")
    for k, v in sorted(conf.items()): print(f"  {k}: {v}")
    if "--dump" in opt:
        for x in body: print(x)


main()
