"""vms_causal_conv1d_fwd_dual next to the two single-direction launches it replaces, at the block's shape (for timing and
for tools/pmc.sh: FETCH_SIZE / WRITE_SIZE of conv_fwd_dual_kernel vs conv_fwd_kernel)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "video-mamba-suite_amd"))
import torch, vms_hip
b, d, L = 8, 1024, 8192
dev, dt = "cuda", torch.bfloat16
xz = torch.randn(b, 2 * d, L, device=dev, dtype=dt); x = xz[:, :d]
w, wb, cb, cbb = torch.randn(d, 4, device=dev), torch.randn(d, 4, device=dev), torch.randn(d, device=dev), torch.randn(d, device=dev)
o1, o2 = torch.empty(b, d, L, device=dev, dtype=dt), torch.empty(b, d, L, device=dev, dtype=dt)
def t(f, n=30):
    for _ in range(10): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1000
def two():
    vms_hip.conv_fwd(x, w, cb, o1, True); vms_hip.conv_fwd(x, wb, cbb, o2, True, reverse=True)
print("two launches %.1f us   one pass %.1f us" % (t(two), t(lambda: vms_hip.conv_fwd_dual(x, w, cb, o1, wb, cbb, o2, True))))
