"""Forward + backward of a module as ONE HIP graph (an extension; the reference has nothing like it).

Small blocks are bound by the host, not by the GPU: the DBM block at BASELINE configs[3] (2, 2304, 512) launches ~35 kernels in
0.42 ms of GPU time but needs 0.75-0.9 ms of Python / autograd time per step.  Every kernel of this package is capture-safe (no host
synchronisation, allocations only through torch's caching allocator, launch attributes set once per device at warm-up), so the
whole step can be recorded once with torch.cuda.CUDAGraph (= hipGraph on ROCm) and replayed with one launch.

    step = GraphedStep(block, example_input)          # warm-up + capture
    out, dx = step(x, grad_out)                        # copies x / grad_out into the captured buffers, replays
    # parameter gradients: p.grad of every parameter of `block` that the forward reaches -- static tensors that the next
    # replay OVERWRITES (no accumulation across replays); replay() re-binds p.grad to them when something
    # (optimizer.zero_grad(set_to_none=True), a clipping routine that assigns .grad) detached them

Data parallel (the reference wraps its models in DistributedDataParallel / nn.DataParallel: run_class_finetuning.py:570-582,
temporal-action-localization/train_eval.py:76): DDP's autograd hooks are host callbacks and cannot be replayed, so under
torch.distributed the BARE module is captured and the gradient exchange is ONE all-reduce of a flat buffer per replay:

    step = GraphedStep(block, example_input, process_group=dist.group.WORLD)

every p.grad is then a view of `step.flat[dtype]`, averaged over the ranks as DDP leaves it.  `allreduce="after"` (default) issues
the collective behind the replay on the replay's stream; `allreduce="captured"` records it inside the graph (RCCL collectives are
capturable; one launch per step in total).  Parameters are broadcast from rank 0 at construction, as DDP does.
"""
import torch


class GraphedStep:
    def __init__(self, module, example_input, autocast_dtype=torch.bfloat16, warmup=3, process_group=None,
                 allreduce="after"):
        assert example_input.is_cuda, "GraphedStep captures GPU work"
        assert allreduce in ("after", "captured")
        self.module = module
        self.x = example_input.detach().clone().requires_grad_()
        self.autocast_dtype = autocast_dtype
        self.pg = process_group
        self.allreduce = allreduce
        self.world = 1
        if process_group is not None:
            import torch.distributed as dist
            self.world = dist.get_world_size(process_group)
            for t in list(module.parameters()) + list(module.buffers()):   # same start on every rank (DDP's init broadcast)
                dist.broadcast(t.data, src=dist.get_global_rank(process_group, 0), group=process_group)
        cand = [p for p in module.parameters() if p.requires_grad]

        def run(gout, params):
            with torch.autocast("cuda", dtype=autocast_dtype or torch.bfloat16, enabled=autocast_dtype is not None):
                y = module(self.x)
            g = gout if gout is not None else torch.zeros_like(y)
            return (y,) + torch.autograd.grad(y, [self.x] + params, g, allow_unused=True)

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):   # kernel attributes, autotuning and allocator pools settle outside the capture
                outs = run(None, cand)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # parameters the forward does not reach (frozen branches, unused variants) get no gradient, as under plain autograd
        self.params = [p for p, g in zip(cand, outs[2:]) if g is not None]
        self.gout = torch.zeros_like(outs[0])
        del outs
        # data parallel: one flat buffer per gradient dtype; the captured step copies the gradients into its views
        self.flat, views = {}, None
        if self.pg is not None:
            sizes = {}
            for p in self.params:
                sizes[p.dtype] = sizes.get(p.dtype, 0) + p.numel()
            self.flat = {dt: torch.zeros(n, device=self.x.device, dtype=dt) for dt, n in sizes.items()}
            offs, views = {dt: 0 for dt in sizes}, []
            for p in self.params:
                views.append(self.flat[p.dtype][offs[p.dtype]:offs[p.dtype] + p.numel()].view_as(p))
                offs[p.dtype] += p.numel()
        self.graph = torch.cuda.CUDAGraph()
        # Under a process group the capture checks only THIS thread's calls: ProcessGroupNCCL's watchdog thread polls the events of earlier
        # (eager) collectives with hipEventQuery, which a capture in the default "global" mode counts as an unsafe call -- the capture is
        # invalidated, the next library call inside it fails, and the watchdog aborts the process (seen as an intermittent SIGABRT of
        # tests/ddp_nccl_worker.py on a cold box, the hipBLASLt "will attempt to recover" warning in front of it).
        mode = {"capture_error_mode": "thread_local"} if self.pg is not None else {}
        with torch.cuda.graph(self.graph, **mode):
            outs = run(self.gout, self.params)
            if views is not None:
                torch._foreach_copy_(views, list(outs[2:]))
                if self.allreduce == "captured":
                    self._exchange()
        self.out, self.dx = outs[0], outs[1]
        self.dparams = tuple(views) if views is not None else outs[2:]
        self._bind()
        if self.pg is not None and self.allreduce == "after":
            self._exchange()   # the capture did not execute anything: nothing to average yet, but RCCL sets up its channels here

    def _exchange(self):
        import torch.distributed as dist
        for buf in self.flat.values():
            if dist.get_backend(self.pg) == "nccl":
                dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.pg)
            else:
                dist.all_reduce(buf, group=self.pg)
                buf.div_(self.world)

    def _bind(self):
        for p, g in zip(self.params, self.dparams):
            if p.grad is not g:
                p.grad = g

    def replay(self):
        """one launch (+ one all-reduce per gradient dtype under data parallel): out, dx and every p.grad are recomputed from
        what self.x / self.gout hold now"""
        self.graph.replay()
        if self.pg is not None and self.allreduce == "after":
            self._exchange()
        self._bind()

    def __call__(self, x, grad_out):
        self.x.data.copy_(x)
        self.gout.copy_(grad_out)
        self.replay()
        return self.out, self.dx
