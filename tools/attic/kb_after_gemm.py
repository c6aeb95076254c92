"""Is the scan slower inside the block step because of what runs before it (clocks / power after ~1.6 ms of GEMMs)
or because of its own operands?  Times the backward / forward scan with events right after a burst of bf16 GEMMs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-mamba-suite_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from kb_dual import problem, bwd, fwd

def main():
    p = problem(0)
    a = torch.randn(65536, 1024, device="cuda", dtype=torch.bfloat16); w = torch.randn(2048, 1024, device="cuda", dtype=torch.bfloat16)
    for name, op in (("bwd", bwd), ("fwd", fwd)):
        for gemms in (0, 4, 12):
            ts = []
            for it in range(30):
                for _ in range(gemms): torch.nn.functional.linear(a, w)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); op(p, False); e1.record()
                if it >= 10: ts.append((e0, e1))
            torch.cuda.synchronize()
            t = sorted(x.elapsed_time(y) for x, y in ts)
            print(f"scan_{name} after {gemms:2d} in_proj-sized GEMMs: median {t[len(t)//2]*1e3:7.1f} us  min {t[0]*1e3:7.1f} us")

if __name__ == "__main__":
    main()
