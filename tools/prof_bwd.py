"""VMS_PROF build only: dump the s_memtime timeline the bwd kernel leaves in the tail of dC."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-mamba-suite_amd")); sys.path.insert(0, ROOT)
import torch, selective_scan_cuda
b, d, L, N = 8, 1024, 8192, 16
dev, dt = "cuda", torch.bfloat16
torch.manual_seed(0)
xz = torch.randn(b, 2 * d, L, device=dev, dtype=dt); u, z = xz[:, :d], xz[:, d:]
delta = (0.5 * torch.rand(d, b, L, device=dev)).to(dt).permute(1, 0, 2)
A = -torch.arange(1, N + 1, device=dev, dtype=torch.float32).repeat(d, 1).contiguous()
B = torch.randn(b, 1, N, L, device=dev, dtype=dt); C = torch.randn(b, 1, N, L, device=dev, dtype=dt)
D = torch.ones(d, device=dev); bias = torch.randn(d, device=dev) - 4.0
out, x, out_z = selective_scan_cuda.fwd(u, delta, A, B, C, D, z, bias, True)
dout = torch.randn(b, d, L, device=dev, dtype=dt); dxz = torch.empty_like(xz)
import vms_hip
# call the C ABI directly so that dC stays fp32 and un-cast
du = torch.empty_like(u); dd = torch.empty_like(delta); dA = torch.zeros_like(A)
dB = torch.zeros(b, 1, N, L, device=dev); dC = torch.zeros(b, 1, N, L, device=dev)
dD = torch.zeros_like(D); db = torch.zeros_like(bias); oz = torch.empty_like(out)
for _ in range(2):
    vms_hip.scan_bwd(u, delta, A, B, C, D, z, bias, dout, x, out, oz, du, dd, dA, dB, dC, dD, db, dxz[:, d:], True)
torch.cuda.synchronize()
raw = dC.flatten().view(torch.int32).cpu().numpy().astype(np.uint32)
per_state, grp = 5, 3
for w in range(2):
    t = raw[(1 << 19) + w * (1 << 17):][: 64 * (3 + 16 * per_state + 4 * grp)]
    t = (t - t[0]).astype(np.int64) & 0xffffffff
    n_per_chunk = 3 + 16 * per_state + 4 * grp
    t = t.reshape(64, n_per_chunk)
    # layout per chunk: [0 top][1 prologue] then per state 5 stamps, after every 4th state 3 group stamps, [2 states done]
    dur = {"prologue": [], "state:wait+widen(s0)": [], "state:local(s1)": [], "state:rowscan(s2)": [], "state:passB+mfma(s3)": [],
           "state:slab(s4)": [], "grp:barrier1": [], "grp:reduce": [], "grp:barrier2": [], "epilogue+next top": []}
    for c in range(64):
        row = t[c]; i = 0
        dur["prologue"].append(row[1] - row[0]); prev = row[1]; i = 2
        for n in range(16):
            for k, name in enumerate(["state:wait+widen(s0)", "state:local(s1)", "state:rowscan(s2)", "state:passB+mfma(s3)", "state:slab(s4)"]):
                dur[name].append(row[i] - prev); prev = row[i]; i += 1
            if n % 4 == 3:
                for name in ["grp:barrier1", "grp:reduce", "grp:barrier2"]:
                    dur[name].append(row[i] - prev); prev = row[i]; i += 1
        if c + 1 < 64:
            dur["epilogue+next top"].append(t[c + 1][0] - prev)
    tot = t[-1][-1] - t[0][0]
    print(f"wave sel {w}: total {tot} ticks")
    for k, v in dur.items():
        v = np.array(v)
        print(f"  {k:26s} n={len(v):5d} mean {v.mean():9.1f} p50 {np.median(v):8.0f} p90 {np.percentile(v,90):8.0f} sum {v.sum()/tot*100:5.1f}%")
