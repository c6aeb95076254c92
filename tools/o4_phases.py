"""Shader-clock time per phase of the 128-VGPR backward scan (a -DVMS_O4_PROF build of the library, tools/variant.sh o4prof -DVMS_O4_PROF;
VMS_HIP_LIB=tools/build/libvms_o4prof.so python tools/o4_phases.py): per wave sums over all chunks and states, averaged over waves."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-mamba-suite_amd")); sys.path.insert(0, ROOT)
import torch
b, d, L, N = [int(x) for x in os.environ.get("KB_SHAPE", "8,1024,8192,16").split(",")]
prof = torch.zeros(8 * 8192, dtype=torch.int64, device="cuda")
os.environ["VMS_O4_PROF_PTR"] = str(prof.data_ptr())
import selective_scan_cuda, vms_hip
dt = torch.bfloat16
torch.manual_seed(0)
xz = torch.randn(b, 2 * d, L, device="cuda", dtype=dt); u, z = xz[:, :d], xz[:, d:]
delta = (0.5 * torch.rand(d, b, L, device="cuda")).to(dt).permute(1, 0, 2)
A = -torch.arange(1, N + 1, device="cuda", dtype=torch.float32).repeat(d, 1).contiguous()
B = torch.randn(b, 1, N, L, device="cuda", dtype=dt); C = torch.randn(b, 1, N, L, device="cuda", dtype=dt)
D = torch.ones(d, device="cuda"); bias = torch.randn(d, device="cuda") - 4.0
out, x, _ = selective_scan_cuda.fwd(u, delta, A, B, C, D, z, bias, True)
A2 = A * 1.1
out2, x2, _ = selective_scan_cuda.fwd(u, delta, A2, B, C, D, z, bias, True, reverse=True)
dout = torch.randn(b, d, L, device="cuda", dtype=dt); dz = torch.empty_like(xz)[:, d:]
da, db_ = (u, delta, A, B, C, D, bias, x, out), (u, delta, A2, B, C, D, bias, x2, out2)
for _ in range(20): selective_scan_cuda.bwd_dual(da, db_, z, dout, dz, True, keep_fp32=True)
torch.cuda.synchronize()
print(vms_hip.lib().vms_last_kernel().decode())
p = prof.cpu().view(-1, 8).double()
p = p[p.sum(1) > 0]
names = ["B/C + exp + adjoint chains + row scan", "state recurrence + products + LDS writes + dA", "workgroup sum of previous state + atomic",
         "4-row sum through LDS + slab write", "barrier", "prologue (+ wait for the row data)", "loop overhead", "epilogue + staging commit + barrier"]
tot = p.sum(1).mean()
n_st = 16 * ((L + 127) // 128)
print(f"waves {p.shape[0]}, cycles per wave {tot:,.0f}")
for k, nm in enumerate(names):
    m = p[:, k].mean().item()
    per = m / n_st if k < 5 else m / ((L + 127) // 128)
    print(f"  {nm:50s} {100 * m / tot:5.1f} %   {per:9.0f} cycles per {'state' if k < 5 else 'chunk'}   (min {p[:, k].min().item() / (n_st if k < 5 else n_st / 16):.0f} max {p[:, k].max().item() / (n_st if k < 5 else n_st / 16):.0f})")
