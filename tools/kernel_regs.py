"""tools/kernel_regs.py <file.s> [filter]: registers / spills / scratch of every kernel in a hipcc -S --cuda-device-only listing."""
import re, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    if flt in name:
        print(f"{name[:110]:110s} agpr {blk.split()[0]:>3s} vgpr {g('vgpr_count'):>3s} vspill {g('vgpr_spill_count'):>3s} sspill {g('sgpr_spill_count'):>3s} scratch {g('private_segment_fixed_size'):>4s} lds {g('group_segment_fixed_size')}")
