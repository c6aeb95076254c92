"""Forward + backward of a module as ONE HIP graph (an extension; the reference has nothing like it).

Small blocks are bound by the host, not by the GPU: the DBM block at BASELINE configs[3] (2, 2304, 512) launches ~35 kernels in
0.42 ms of GPU time but needs 0.75-0.9 ms of Python / autograd time per step.  Every kernel of this package is capture-safe (no host
synchronisation, allocations only through torch's caching allocator, launch attributes set at warm-up), so the whole step can be
recorded once with torch.cuda.CUDAGraph (= hipGraph on ROCm) and replayed with one launch.

    step = GraphedStep(block, example_input)          # warm-up + capture
    out, dx = step(x, grad_out)                        # copies x / grad_out into the captured buffers, replays
    # parameter gradients: p.grad of every parameter of `block` (static tensors, overwritten by the next replay)

Single-process use only: DistributedDataParallel's gradient hooks are not part of the captured work.
"""
import torch


class GraphedStep:
    def __init__(self, module, example_input, autocast_dtype=torch.bfloat16, warmup=3):
        assert example_input.is_cuda, "GraphedStep captures GPU work"
        self.module = module
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.x = example_input.detach().clone().requires_grad_()
        self.autocast_dtype = autocast_dtype

        def run(gout):
            with torch.autocast("cuda", dtype=autocast_dtype or torch.bfloat16, enabled=autocast_dtype is not None):
                y = module(self.x)
            g = gout if gout is not None else torch.zeros_like(y)
            return (y,) + torch.autograd.grad(y, [self.x] + self.params, g)

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):   # kernel attributes, autotuning and allocator pools settle outside the capture
                y = run(None)[0]
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.gout = torch.zeros_like(y)
        del y
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            outs = run(self.gout)
        self.out, self.dx, self.dparams = outs[0], outs[1], outs[2:]
        for p, g in zip(self.params, self.dparams):
            p.grad = g

    def replay(self):
        """one launch: out, dx and every p.grad are recomputed from what self.x / self.gout hold now"""
        self.graph.replay()

    def __call__(self, x, grad_out):
        self.x.data.copy_(x)
        self.gout.copy_(grad_out)
        self.graph.replay()
        return self.out, self.dx
