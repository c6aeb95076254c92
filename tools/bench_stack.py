"""BASELINE configs[2]-like secondary line: a 12-layer stack of Block(Add -> RMSNorm -> ViM mixer) with the fused
add+norm kernels, d_model 768, expand 1, (B, L) = (8, 3136), bf16 autocast, forward + backward, eager and as one HIP graph."""
import os, sys, functools, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "video-mamba-suite_amd"))
from mamba_ssm.modules.mamba_simple import Mamba, Block
from mamba_ssm.ops.triton.layernorm import RMSNorm, rms_norm_fn

class Stack(torch.nn.Module):
    def __init__(self, depth, d_model):
        super().__init__()
        mixer = functools.partial(Mamba, expand=1, bimamba_type="v2")
        self.layers = torch.nn.ModuleList([Block(d_model, mixer, norm_cls=RMSNorm, fused_add_norm=True, residual_in_fp32=True)
                                           for _ in range(depth)])
        self.norm_f = RMSNorm(d_model)
    def forward(self, x):
        res = None
        for layer in self.layers:
            x, res = layer(x, res)
        return rms_norm_fn(x, self.norm_f.weight, self.norm_f.bias, residual=res, eps=self.norm_f.eps, prenorm=False,
                           residual_in_fp32=True)

def main():
    depth, d_model, B, L = 12, 768, 8, 3136
    torch.manual_seed(0)
    net = Stack(depth, d_model).cuda()
    params = list(net.parameters())
    x = torch.randn(B, L, d_model, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    g = torch.randn(B, L, d_model, device="cuda", dtype=torch.bfloat16)
    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = net(x)
        return torch.autograd.grad(y, [x] + params, g.to(y.dtype))
    def timed(fn, n=10, w=5):
        for _ in range(w): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    eager = timed(step)
    outs = step()
    assert all(torch.isfinite(t.float()).all() for t in outs), "non-finite gradient"
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    replay = timed(graph.replay)
    print(f"{depth} x Block(RMSNorm + ViM d_model {d_model}) at (B, L) = ({B}, {L}): eager {eager:.2f} ms = {B * L / eager / 1e3:.2f} M tok/s, "
          f"one HIP graph {replay:.2f} ms = {B * L / replay / 1e3:.2f} M tok/s, peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")

if __name__ == "__main__":
    main()
