"""tools/guard_probe.py one level down: the C-ABI entry points through the ctypes wrappers of vms_hip, which take EVERY buffer from the
caller -- so outputs, accumulators and checkpoint buffers can be guarded too (an over-WRITE is the worse bug).  Per (entry point, shape)
one subprocess walks the operands, placing one at a time so that its last byte ends its own 10 MiB hipMalloc segment; it prints the
operand's name before each launch, so a GPU memory fault names the culprit.
    python tools/guard_probe_abi.py [filter]     # all (or the matching) cases
    python tools/guard_probe_abi.py --one <case> # one case in this process"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-mamba-suite_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
SEG = 10 << 20


def at_end(t):
    import torch
    span = (sum((s - 1) * st for s, st in zip(t.shape, t.stride())) + 1) * t.element_size()
    seg = max(SEG, -(-span // (2 << 20)) * (2 << 20))     # >= 10 MB requests get their own segment, sized in 2 MiB units
    big = torch.empty(seg, dtype=torch.uint8, device=t.device)
    v = big[seg - span:].view(t.dtype).as_strided(t.shape, t.stride())
    v.copy_(t)
    v._keep = big
    return v


def cases():
    """name -> (dict of operand tensors, fn(dict))"""
    import torch
    import torch.nn.functional as F
    import vms_hip as V
    dev, out = "cuda", {}
    bf, f32 = torch.bfloat16, torch.float32
    R = lambda *s, dt=bf: torch.randn(*s, device=dev).to(dt)
    Z = lambda *s, dt=f32: torch.zeros(*s, device=dev, dtype=dt)

    # ---- scans: shapes that select the generic, ragged, LDS / 4-rows-per-wave, short-row, split and state-group kernels
    for (b, d, L, N, dt) in ((2, 32, 100, 8, bf), (3, 128, 8, 16, bf), (2, 64, 24, 16, bf), (2, 64, 1040, 16, bf), (4, 512, 64, 16, bf), (64, 64, 8, 16, bf),
                             (1, 64, 8192, 16, bf), (1, 768, 4096, 16, bf), (2, 64, 1040, 16, f32), (1, 32, 13, 16, torch.float16)):
        for rev in (False, True):
            pad = (-L) % 16
            B0, C0 = R(b, 1, N, L, dt=dt), R(b, 1, N, L, dt=dt)
            Bp = F.pad(B0, (pad, 0) if rev else (0, pad)); Cp = F.pad(C0, (pad, 0) if rev else (0, pad))
            T = dict(u=R(b, d, L, dt=dt), delta=(0.5 * torch.rand(d, b, L, device=dev)).to(dt).permute(1, 0, 2), A=-torch.rand(d, N, device=dev) - 0.1,
                     Bp=Bp, Cp=Cp, D=R(d, dt=f32), z=R(b, d, L, dt=dt), bias=torch.rand(d, device=dev), out=R(b, d, L, dt=dt), out_z=R(b, d, L, dt=dt),
                     dout=R(b, d, L, dt=dt), du=R(b, d, L, dt=dt), ddelta=R(b, d, L, dt=dt), dz=R(b, d, L, dt=dt),
                     dA=Z(d, N), dB=Z(b, 1, N, L), dC=Z(b, 1, N, L), dD=Z(d), dbias=Z(d))
            view = (lambda t, pad=pad: t[..., pad:]) if rev else (lambda t, L=L: t[..., :L])
            # x as the library lays it out for this problem: learn the geometry once, then guard the WIDE buffer behind the view
            x0 = V.scan_fwd(T["u"], T["delta"], T["A"], view(Bp), view(Cp), T["D"], T["z"], T["bias"], T["out"], T["out_z"], None, True, reverse=rev, bc_pad=pad)
            pitch = x0.stride(2)
            T["xwide"] = torch.zeros(b, d, x0.shape[2], pitch, device=dev)

            def fwd(t, rev=rev, view=view, pad=pad, N=N):
                V.scan_fwd(t["u"], t["delta"], t["A"], view(t["Bp"]), view(t["Cp"]), t["D"], t["z"], t["bias"], t["out"], t["out_z"], t["xwide"][..., :2 * N],
                           True, reverse=rev, bc_pad=pad)

            def bwd(t, rev=rev, view=view, pad=pad, N=N):
                x = t["xwide"][..., :2 * N]
                V.scan_fwd(t["u"], t["delta"], t["A"], view(t["Bp"]), view(t["Cp"]), t["D"], t["z"], t["bias"], t["out"], t["out_z"], x, True, reverse=rev, bc_pad=pad)
                V.scan_bwd(t["u"], t["delta"], t["A"], view(t["Bp"]), view(t["Cp"]), t["D"], t["z"], t["bias"], t["dout"], x, t["out"], None, t["du"], t["ddelta"],
                           t["dA"], t["dB"], t["dC"], t["dD"], t["dbias"], t["dz"], True, reverse=rev, bc_pad=pad)
            tag = f"b{b} d{d} L{L} N{N} {str(dt)[6:]} rev{int(rev)}"
            fw = {k: T[k] for k in ("u", "delta", "A", "Bp", "Cp", "D", "z", "bias", "out", "out_z", "xwide")}
            out["scan_fwd " + tag] = (fw, fwd)
            out["scan_bwd " + tag] = (T, bwd)

    # ---- both directions' backward scans as one call (4- and 8-wave grids)
    for (b, d, L) in ((8, 256, 64), (8, 512, 72)):
        N = 16
        dirs = []
        z, dout = R(b, d, L), R(b, d, L)
        T = dict(z=z, dout=dout, dz=R(b, d, L))
        for i, rev in enumerate((False, True)):
            p = "ab"[i]
            T.update({p + "u": R(b, d, L), p + "delta": (0.5 * torch.rand(d, b, L, device=dev)).to(bf).permute(1, 0, 2), p + "A": -torch.rand(d, N, device=dev) - 0.1,
                      p + "B": R(b, 1, N, L), p + "C": R(b, 1, N, L), p + "D": R(d, dt=f32), p + "bias": torch.rand(d, device=dev), p + "out": R(b, d, L),
                      p + "du": R(b, d, L), p + "ddelta": R(b, d, L), p + "dA": Z(d, N), p + "dB": Z(b, 1, N, L), p + "dC": Z(b, 1, N, L), p + "dD": Z(d), p + "dbias": Z(d)})
            x0 = V.scan_fwd(T[p + "u"], T[p + "delta"], T[p + "A"], T[p + "B"], T[p + "C"], T[p + "D"], z, T[p + "bias"], T[p + "out"], R(b, d, L), None, True, reverse=rev)
            T[p + "xwide"] = torch.zeros(b, d, x0.shape[2], x0.stride(2), device=dev)

        def dual(t, N=N):
            args = []
            for i, rev in enumerate((False, True)):
                p = "ab"[i]
                x = t[p + "xwide"][..., :2 * N]
                V.scan_fwd(t[p + "u"], t[p + "delta"], t[p + "A"], t[p + "B"], t[p + "C"], t[p + "D"], t["z"], t[p + "bias"], t[p + "out"], torch.empty_like(t["z"]), x, True, reverse=rev)
                args.append((t[p + "u"], t[p + "delta"], t[p + "A"], t[p + "B"], t[p + "C"], t[p + "D"], t["z"], t[p + "bias"], t["dout"], x, t[p + "out"], None, t[p + "du"],
                             t[p + "ddelta"], t[p + "dA"], t[p + "dB"], t[p + "dC"], t[p + "dD"], t[p + "dbias"], t["dz"] if i == 0 else None, True, rev))
            V.scan_bwd_dual(*args)
        out[f"scan_bwd_dual b{b} d{d} L{L}"] = (T, dual)

    # ---- conv1d
    for (b, d, L, dt) in ((2, 64, 24, bf), (1, 32, 13, torch.float16), (2, 96, 1040, bf), (3, 40, 777, f32), (2, 256, 4096, bf)):
        for rev in (False, True):
            T = dict(x=R(b, d, L, dt=dt), w=R(d, 4, dt=f32), bias=R(d, dt=f32), out=R(b, d, L, dt=dt), dout=R(b, d, L, dt=dt), dx=R(b, d, L, dt=dt), dw=Z(d, 4), db=Z(d))
            out[f"conv_fwd b{b} d{d} L{L} {str(dt)[6:]} rev{int(rev)}"] = ({k: T[k] for k in ("x", "w", "bias", "out")}, lambda t, rev=rev: V.conv_fwd(t["x"], t["w"], t["bias"], t["out"], True, reverse=rev))
            out[f"conv_bwd b{b} d{d} L{L} {str(dt)[6:]} rev{int(rev)}"] = ({k: T[k] for k in ("x", "w", "bias", "dout", "dx", "dw", "db")},
                                                                            lambda t, rev=rev: V.conv_bwd(t["x"], t["w"], t["bias"], t["dout"], t["dx"], t["dw"], t["db"], True, reverse=rev))
    # channel-last
    xcl = R(2, 700, 64).transpose(1, 2)
    out["conv_fwd channel_last"] = (dict(x=xcl, w=R(64, 4, dt=f32), bias=R(64, dt=f32), out=R(2, 700, 64).transpose(1, 2)), lambda t: V.conv_fwd(t["x"], t["w"], t["bias"], t["out"], True))

    # ---- the block's fused head / tail and small projections, at a block-like and at a short-sequence shape
    for (b, d, L, Rk) in ((2, 128, 200, 8), (3, 256, 8, 16), (2, 768, 3136, 48)):
        N, m = 16, Rk + 32
        T = dict(x=R(b, d, L), cw=R(d, 4, dt=f32), cb=R(d, dt=f32), cw_b=R(d, 4, dt=f32), cb_b=R(d, dt=f32), w_x=R(m, d), w_x_b=R(m, d),
                 out=R(b, d, L), out_b=R(b, d, L), x_dbl=R(b, m, L), x_dbl_b=R(b, m, L))
        if V.conv_xproj_dual_eligible(T["x"], T["cw"], T["cb"], T["cw_b"], T["cb_b"], T["w_x"], T["w_x_b"]):
            out[f"conv_xproj_dual b{b} d{d} L{L} m{m}"] = (T, lambda t: V.conv_xproj_dual(t["x"], t["cw"], t["cb"], t["out"], t["cw_b"], t["cb_b"], t["out_b"], t["w_x"], t["w_x_b"],
                                                                                         t["x_dbl"], t["x_dbl_b"]))
        T = dict(x=R(b, d, L), cw=R(d, 4, dt=f32), cb=R(d, dt=f32), w_x=R(m, d), w_x_b=R(m, d), out=R(b, d, L), out_b=R(b, d, L))
        out[f"conv_fwd_dual b{b} d{d} L{L}"] = ({k: T[k] for k in ("x", "cw", "cb", "out", "out_b")} | dict(cw_b=R(d, 4, dt=f32), cb_b=R(d, dt=f32)),
                                                 lambda t: V.conv_fwd_dual(t["x"], t["cw"], t["cb"], t["out"], t["cw_b"], t["cb_b"], t["out_b"], True))
        # delta = W_dt x_dbl[:R]  and its weight gradient
        T = dict(w=R(d, Rk), inp=R(b, m, L)[:, :Rk], out=R(d, b, L).permute(1, 0, 2))
        out[f"proj_apply b{b} d{d} L{L} R{Rk}"] = (T, lambda t: V.proj_apply(t["w"], t["inp"], t["out"]))
        T = dict(p=R(b, m, L)[:, :Rk], q=R(d, b, L).permute(1, 0, 2), dw=Z(d, Rk))
        out[f"proj_wgrad b{b} d{d} L{L} R{Rk}"] = (T, lambda t: V.proj_wgrad(t["p"], t["q"], t["dw"], transposed=True))
        # x_dbl = W_x conv_out (both directions in one launch), d_dt = W_dt^T ddelta with the dB / dC cast riding along
        T = dict(w=R(m, d), inp=R(b, d, L), out=R(b, m, L), w2=R(m, d), inp2=R(b, d, L), out2=R(b, m, L))
        if V.proj_kred_eligible(T["w"], T["inp"], T["out"]):
            out[f"proj_kred fwd b{b} d{d} L{L} m{m}"] = (T, lambda t: V.proj_kred(t["w"], t["inp"], t["out"], t["w2"], t["inp2"], t["out2"]))
        T = dict(wdt=R(d, Rk), ddelta=R(d, b, L).permute(1, 0, 2), dx_dbl=R(b, m, L), cast=torch.randn(2, b, N, L, device=dev))
        if V.proj_kred_eligible(T["wdt"].t(), T["ddelta"], T["dx_dbl"][:, :Rk]):
            out[f"proj_kred bwd b{b} d{d} L{L} R{Rk}"] = (T, lambda t, Rk=Rk: V.proj_kred(t["wdt"].t(), t["ddelta"], t["dx_dbl"][:, :Rk], cast_src=t["cast"]))
        if Rk + 32 >= 33:
            for rev in (False, True):
                T = dict(x=R(b, d, L), du=R(d, b, L).permute(1, 0, 2), dx_dbl=R(b, m, L), w_x=R(m, d), cw=R(d, 4, dt=f32), cb=R(d, dt=f32), dx=R(b, d, L),
                         dcw=Z(d, 4), dcb=Z(d), dwx=Z(m, d))
                out[f"proj_conv_bwd b{b} d{d} L{L} m{m} rev{int(rev)}"] = (T, lambda t, rev=rev: V.proj_conv_bwd(t["x"], t["du"], t["dx_dbl"], t["w_x"], t["cw"], t["cb"], t["dx"], t["dcw"],
                                                                                                                 t["dcb"], t["dwx"], reverse=rev))

    # ---- fused add + norm
    for (rows, cols, rms) in ((24, 128, False), (25088 // 8, 768, True), (17, 40, False)):
        T = dict(x=R(rows, cols), res=torch.randn(rows, cols, device=dev), w=R(cols, dt=f32), b=R(cols, dt=f32), y=R(rows, cols), res_out=torch.randn(rows, cols, device=dev),
                 mean=Z(rows), rstd=Z(rows))
        out[f"norm_fwd {rows}x{cols} rms{int(rms)}"] = (T, lambda t, rms=rms: V.norm_fwd(t["x"], t["res"], t["w"], None if rms else t["b"], t["y"], t["res_out"], None if rms else t["mean"],
                                                                                            t["rstd"], 1e-5, rms))
        npart = V.norm_bwd_partials(rows, cols)
        T = dict(s=torch.randn(rows, cols, device=dev), dy=R(rows, cols), w=R(cols, dt=f32), mean=Z(rows), rstd=torch.rand(rows, device=dev) + 0.5,
                 dres_out=torch.randn(rows, cols, device=dev), dx=R(rows, cols), dres_in=torch.randn(rows, cols, device=dev), dwp=Z(npart, cols), dbp=Z(npart, cols))
        out[f"norm_bwd {rows}x{cols} rms{int(rms)}"] = (T, lambda t, rms=rms: V.norm_bwd(t["s"], t["dy"], t["w"], None if rms else t["mean"], t["rstd"], t["dres_out"], t["dx"], t["dres_in"],
                                                                                            t["dwp"], None if rms else t["dbp"], rms))

    # ---- single-token steps
    T = dict(state=R(2, 96, 16, dt=f32), x=R(2, 96), dt=R(2, 96), A=-torch.rand(96, 16, device=dev), B=R(2, 16), C=R(2, 16), D=R(96, dt=f32), z=R(2, 96), bias=R(96, dt=f32), out=R(2, 96))
    out["state_update"] = (T, lambda t: V.state_update(t["state"], t["x"], t["dt"], t["A"], t["B"], t["C"], t["D"], t["z"], t["bias"], t["out"], True))
    T = dict(x=R(2, 96), cs=R(2, 96, 4), w=R(96, 4, dt=f32), b=R(96, dt=f32), out=R(2, 96))
    out["conv_update"] = (T, lambda t: V.conv_update(t["x"], t["cs"], t["w"], t["b"], t["out"], True))
    return out


def run_one(name):
    import torch
    T, fn = cases()[name]
    fn(T)                       # unguarded once: a problem the entry point declines raises here, not in a probe
    torch.cuda.synchronize()
    for k in T:
        if not torch.is_tensor(T[k]):
            continue
        print("probing", k, flush=True)
        t2 = dict(T)
        t2[k] = at_end(T[k])
        fn(t2)
        torch.cuda.synchronize()
    print("case ok", flush=True)


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--one":
        return run_one(sys.argv[2])
    import torch  # noqa: F401
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    names = [n for n in cases() if flt in n]
    bad, probes = 0, 0
    for n in names:
        r = subprocess.run([sys.executable, __file__, "--one", n], capture_output=True, text=True)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("probing")]
        probes += len(lines)
        if "case ok" not in r.stdout:
            bad += 1
            err = [ln for ln in (r.stdout + r.stderr).splitlines() if "fault" in ln or "Error" in ln or "error" in ln][-1:] or ["?"]
            print(f"FAULT {n}: at operand '{lines[-1][8:] if lines else '(unguarded run)'}' :: {err[0][:160]}", flush=True)
    print(f"{len(names)} cases, {probes} probes, {bad} faults")


if __name__ == "__main__":
    main()
