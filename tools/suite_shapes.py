"""tools/suite_shapes.py -- every (model, shape) the reference's task code instantiates, one row each:
ms per layer step (forward + backward under autocast bf16), the kernel every C-ABI launch of the step chose (vms_last_kernel()
behind each launch, from the compiled binding's timing records), and a FLAG when a call landed on a generic scan kernel, the
ragged generation, or the unfused tail / head.

  python tools/suite_shapes.py [filter] [--md out.md] [--json out.json]

Shapes (reference files under /root/reference/video-mamba-suite/; nothing of the reference is read at run time):
  ViViM tiny / small   action-recognition/models/vivim.py:406-423, 511-580 (Block = Add -> RMSNorm -> ViM, expand 2; 8 / 16 frames x
                       196 patches + cls tokens: 1569 / 1576 / 3137 / 3152)
  CLIP ViViM           egocentric-understanding/avion/models/model_clip.py:945-967 (ssm_cfg = dict(d_state=4), embed_dim 192, if_devide_out)
  TimeMamba            egocentric-understanding/avion/models/timemamba.py:115-147 (Mamba(768, expand=1): along time '(b n) t d', or joint)
  TAL                  temporal-action-localization/libs/modeling/blocks.py:899-942, backbones.py:282-288 (DBM expand 1 / ViM, 2304 ... 144)
  TAS                  temporal-action-segmentation/modeling/blocks.py:910, 949 (batch 1, thousands of frames, 64 / 256 channels)
  PDVC                 video-dense-captioning/pdvc/deformable_transformer.py:244-246 (d_model 512, 4 feature levels of 100 frames: 188 tokens)
  UniVTG               video-temporal-grounding/model/univtg_mamba.py:55-57 (hidden 256, 75 clips + query tokens)
  LSTR                 action-anticipation/.../models/lstr.py:32, 138 (d_model 1024, work memory 32 / long memory 512 samples)
"""
import json
import os
import statistics
import sys
from functools import partial

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-mamba-suite_amd"))
sys.path.insert(0, ROOT)
import torch

# name, kind (block = Add -> RMSNorm -> mixer with fused add+norm; vim / dbm = the bare mixer), mixer kwargs, (batch, seqlen)
SHAPES = [
    ("vivim_tiny 8f cls-end", "block", dict(d_model=192, expand=2), (8, 1569)),
    ("vivim_tiny 8f cls-per-frame", "block", dict(d_model=192, expand=2), (8, 1576)),
    ("vivim_small 16f cls-end", "block", dict(d_model=384, expand=2), (8, 3137)),
    ("vivim_small 16f cls-per-frame", "block", dict(d_model=384, expand=2), (8, 3152)),
    ("clip_vivim d_state=4, 4f", "block", dict(d_model=192, expand=2, d_state=4, if_devide_out=True), (32, 785)),
    ("clip_vivim d_state=4, 16f", "block", dict(d_model=192, expand=2, d_state=4, if_devide_out=True), (8, 3137)),
    ("timemamba time 4f", "vim", dict(d_model=768, expand=1), (1568, 4)),
    ("timemamba time 8f", "vim", dict(d_model=768, expand=1), (1568, 8)),
    ("timemamba time 16f", "vim", dict(d_model=768, expand=1), (1568, 16)),
    ("timemamba joint 8f", "vim", dict(d_model=768, expand=1), (8, 1568)),
    ("timemamba joint 16f", "vim", dict(d_model=768, expand=1), (8, 3136)),
    ("tal dbm 2304", "dbm", dict(d_model=512, expand=1), (2, 2304)),
    ("tal dbm 1152", "dbm", dict(d_model=512, expand=1), (2, 1152)),
    ("tal dbm 576", "dbm", dict(d_model=512, expand=1), (2, 576)),
    ("tal dbm 288", "dbm", dict(d_model=512, expand=1), (2, 288)),
    ("tal dbm 144", "dbm", dict(d_model=512, expand=1), (2, 144)),
    ("tal vim 2304", "vim", dict(d_model=512, expand=2), (2, 2304)),
    ("tas vim d64", "vim", dict(d_model=64, expand=2), (1, 6000)),
    ("tas dbm d64", "dbm", dict(d_model=64, expand=1), (1, 6000)),
    ("tas vim d256", "vim", dict(d_model=256, expand=2), (1, 6000)),
    ("pdvc vim 188", "vim", dict(d_model=512, expand=2), (1, 188)),
    ("pdvc dbm 188", "dbm", dict(d_model=512, expand=1), (1, 188)),
    ("univtg vim 107", "vim", dict(d_model=256, expand=2), (32, 107)),
    ("univtg dbm 107", "dbm", dict(d_model=256, expand=1), (32, 107)),
    ("lstr work 32", "vim", dict(d_model=1024, expand=2), (16, 32)),
    ("lstr long 512", "vim", dict(d_model=1024, expand=2), (16, 512)),
]


def build(kind, kw, dev):
    from mamba_ssm.modules.mamba_simple import Block, Mamba
    from mamba_ssm.modules.mamba_new import Mamba as DBM
    from mamba_ssm.ops.triton.layernorm import RMSNorm
    kw = dict(kw)
    d_model = kw.pop("d_model")
    if kind == "dbm":
        return DBM(d_model, d_conv=4, **kw).to(dev)
    mixer = partial(Mamba, d_conv=4, bimamba_type="v2", **kw)
    if kind == "vim":
        return mixer(d_model).to(dev)
    return Block(d_model, mixer, norm_cls=partial(RMSNorm, eps=1e-5), fused_add_norm=True, residual_in_fp32=True).to(dev)


def flags_of(records):
    fl = []
    entries = [e for e, _, _ in records]
    for e, _, k in records:
        if "scan" in e and "generic" in k:
            fl.append(f"GENERIC {k}")
        if "ragged" in k:
            fl.append(f"RAGGED {k}")
    if "vms_proj_conv_bwd" not in entries:
        fl.append("UNFUSED TAIL")
    if "vms_conv_xproj_dual" not in entries and not any(k.startswith("proj_kred") for _, _, k in records):
        fl.append("UNFUSED HEAD")
    return sorted(set(fl))


def run_one(name, kind, kw, shape, dev="cuda"):
    import vms_hip
    assert vms_hip.ext() is not None, "the compiled binding (_vms_torch.so) is needed for the per-launch kernel names"
    torch.manual_seed(0)
    m = build(kind, kw, dev)
    b, l = shape
    d_model = kw["d_model"]
    h = torch.randn(b, l, d_model, device=dev, dtype=torch.bfloat16, requires_grad=True)
    res = torch.randn(b, l, d_model, device=dev, dtype=torch.float32) if kind == "block" else None
    g = torch.randn(b, l, d_model, device=dev, dtype=torch.bfloat16)

    def step():
        m.zero_grad(set_to_none=True)
        h.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = m(h, res) if kind == "block" else m(h)
        if kind == "block":
            torch.autograd.backward([out[0], out[1]], [g, torch.zeros_like(out[1])])
        else:
            out.backward(g)

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(15):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        step()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    vms_hip.start_timing(reserve=64)
    step()
    torch.cuda.synchronize()
    vms_hip._timing = None
    recs = [(e, ms, k) for e, ms, k in vms_hip.ext().timing_stop_detail()]
    gpu_ms = sum(ms for _, ms, _ in recs)
    kern = {}
    for e, ms, k in recs:
        kern.setdefault(e.replace("vms_", ""), []).append((k, round(ms * 1e3, 1)))
    return {"name": name, "kind": kind, "mixer": kw, "batch": b, "seqlen": l, "ms_per_step": statistics.median(ts),
            "vms_kernel_ms": gpu_ms, "launches": kern, "flags": flags_of(recs)}


def main():
    args = [a for a in sys.argv[1:]]
    md = js = None
    if "--md" in args:
        md = args[args.index("--md") + 1]
        del args[args.index("--md"):args.index("--md") + 2]
    if "--json" in args:
        js = args[args.index("--json") + 1]
        del args[args.index("--json"):args.index("--json") + 2]
    flt = args[0] if args else ""
    rows = []
    for name, kind, kw, shape in SHAPES:
        if flt and flt not in name:
            continue
        try:
            r = run_one(name, kind, kw, shape)
        except Exception as e:  # noqa: BLE001
            r = {"name": name, "kind": kind, "mixer": kw, "batch": shape[0], "seqlen": shape[1], "error": f"{type(e).__name__}: {e}"[:300]}
        rows.append(r)
        print(json.dumps(r), flush=True)
        torch.cuda.empty_cache()
    lines = ["| model, shape (B, L) | mixer | ms / layer step (eager) | scan fwd | scan bwd | head / tail / projections | flags |", "|---|---|---|---|---|---|---|"]
    for r in rows:
        mx = ", ".join(f"{k}={v}" for k, v in r["mixer"].items())
        if "error" in r:
            lines.append(f"| {r['name']} ({r['batch']}, {r['seqlen']}) | {r['kind']} {mx} | ERROR {r['error']} | | | | |")
            continue
        ln = r["launches"]
        fmt = lambda keys: "; ".join(f"{k} {us:.0f}" for key in keys for k, us in ln.get(key, [])) or "-"
        other = [k for k in ln if "scan" not in k and "norm" not in k and k != "param_prep"]
        lines.append(f"| {r['name']} ({r['batch']}, {r['seqlen']}) | {r['kind']} {mx} | {r['ms_per_step']:.3f} (kernels of this library {r['vms_kernel_ms']:.3f}) | "
                     f"{fmt(['selective_scan_fwd'])} | {fmt(['selective_scan_bwd', 'selective_scan_bwd_dual'])} | {fmt(other)} | {', '.join(r['flags']) or 'ok'} |")
    text = "\n".join(lines)
    print(text)
    if md:
        open(md, "w").write(text + "\n")
    if js:
        json.dump(rows, open(js, "w"), indent=1)


if __name__ == "__main__":
    main()
