"""gfx950 wants one wait state between a transcendental's write (v_exp / v_rcp / v_log / v_sqrt / v_rsq / v_sin / v_cos) and a
non-transcendental VALU read of the result.  The compiler inserts it for its own instructions, but it does not look inside an
asm statement: a `v_pk_fma_f32` written as inline asm directly behind the `v_exp_f32` that feeds it reads the register's OLD
value (round 4: the first version of the backward carry pass -- 28 such pairs, 15 of 16 states wrong).

This scans the gfx950 ISA of the sources that hold asm VALU statements for that pattern.
usage: python tools/isa_hazards.py [file.s ...]      (no arguments: compile csrc/selective_scan_{fwd,bwd}_pair.hip to ISA first)
exit status 1 if a hazard was found."""
import concurrent.futures
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CSRC = os.path.join(ROOT, "video-mamba-suite_amd", "csrc")
ASM_SOURCES = ("selective_scan_fwd_pair.hip", "selective_scan_bwd_pair.hip")     # grep 'asm("v_' csrc/*.hip
TRANS = re.compile(r"^\s*v_(exp|rcp|log|sqrt|rsq|sin|cos)_(f32|f16|legacy_f32)\S*\s+(v\d+)")
ASM_VALU = ("v_pk_fma_f32",)                                                     # the VALU instructions written as inline asm


def scan(path):
    """[(trans dst, consumer line)] where an ASM_VALU instruction directly follows the transcendental that writes one of its sources"""
    found, prev = [], None
    for line in open(path):
        s = line.strip()
        if not s or s[0] in ";." or s.endswith(":"):
            continue
        if prev is not None and s.startswith(ASM_VALU):
            reg = int(prev[1:])
            srcs = s.split(None, 1)[1].split(",", 1)[1]
            if any(int(a) <= reg <= int(b) for a, b in re.findall(r"v\[(\d+):(\d+)\]", srcs)):
                found.append((prev, s))
        m = TRANS.match(line)
        prev = m.group(3) if m else None
    return found


def to_isa(src, out):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S",
                           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, os.path.join(CSRC, src), "-o", out],
                          stderr=subprocess.DEVNULL)
    return out


def main(argv):
    if argv:
        files = argv
    else:
        tmp = tempfile.mkdtemp(prefix="vms_isa_")
        with concurrent.futures.ThreadPoolExecutor(len(ASM_SOURCES)) as ex:
            files = list(ex.map(lambda s: to_isa(s, os.path.join(tmp, s[:-4] + ".s")), ASM_SOURCES))
    bad = 0
    for f in files:
        h = scan(f)
        bad += len(h)
        for reg, use in h[:8]:
            print(f"{os.path.basename(f)}: {reg} written by a transcendental, read by the next instruction: {use}")
        print(f"{os.path.basename(f)}: {len(h)} hazard(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
