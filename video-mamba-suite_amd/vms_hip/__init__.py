"""ctypes binding of libvms_hip.so (C ABI: include/vms_hip.h).

PyTorch is used only as plumbing here: device memory (`data_ptr()`), element strides and the
current HIP stream.  There is NO fallback: if the library is missing, or a tensor is not on a
GPU, the call raises (the reference's extensions behave the same way: import error at module
import, TORCH_CHECK(x.is_cuda()) at call time -- selective_scan.cpp:246-250).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_DEFAULT_LIB = os.path.join(_HERE, "libvms_hip.so")
LIB_PATH = os.environ.get("VMS_HIP_LIB", _DEFAULT_LIB)  # env: A/B builds in tools/

VMS_F32, VMS_F16, VMS_BF16 = 0, 1, 2
_DTYPE = {torch.float32: VMS_F32, torch.float16: VMS_F16, torch.bfloat16: VMS_BF16}

_i32, _i64, _vp, _fp = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p


class ScanFwdParams(ctypes.Structure):
    _fields_ = (
        [(n, _i32) for n in ("batch", "dim", "seqlen", "dstate", "n_groups", "n_chunks", "dtype",
                             "is_variable_B", "is_variable_C", "delta_softplus")]
        + [(n, _vp) for n in ("u", "delta", "A", "B", "C", "D", "z", "delta_bias", "out", "out_z", "x")]
        + [(n, _i64) for n in (
            "u_batch_stride", "u_d_stride", "delta_batch_stride", "delta_d_stride",
            "z_batch_stride", "z_d_stride", "out_batch_stride", "out_d_stride",
            "out_z_batch_stride", "out_z_d_stride", "A_d_stride", "A_dstate_stride",
            "B_batch_stride", "B_group_stride", "B_d_stride", "B_dstate_stride",
            "C_batch_stride", "C_group_stride", "C_d_stride", "C_dstate_stride", "x_chunk_stride")]
        + [("x_has_sub", _i32), ("reverse", _i32), ("out_z_accumulate", _i32), ("bc_pad", _i32),
           ("workspace", _vp), ("workspace_bytes", _i64), ("impl", _i32), ("segments", _i32),
           ("reverse_from", _i32), ("is_complex", _i32)]
    )


class ScanBwdParams(ctypes.Structure):
    _fields_ = (
        [("f", ScanFwdParams), ("dout", _vp), ("du", _vp), ("ddelta", _vp), ("dz", _vp),
         ("dA", _fp), ("dB", _fp), ("dC", _fp), ("dD", _fp), ("ddelta_bias", _fp)]
        + [(n, _i64) for n in (
            "dout_batch_stride", "dout_d_stride", "du_batch_stride", "du_d_stride",
            "ddelta_batch_stride", "ddelta_d_stride", "dz_batch_stride", "dz_d_stride",
            "dA_d_stride", "dA_dstate_stride",
            "dB_batch_stride", "dB_group_stride", "dB_d_stride", "dB_dstate_stride",
            "dC_batch_stride", "dC_group_stride", "dC_d_stride", "dC_dstate_stride")]
        + [("dz_accumulate", _i32), ("reserved0", _i32)]
    )


class ConvFwdParams(ctypes.Structure):
    _fields_ = (
        [(n, _i32) for n in ("batch", "dim", "seqlen", "width", "dtype", "wdtype", "silu_activation", "reverse")]
        + [(n, _vp) for n in ("x", "weight", "bias", "out")]
        + [(n, _i64) for n in ("x_batch_stride", "x_c_stride", "x_l_stride", "weight_c_stride",
                               "weight_width_stride", "out_batch_stride", "out_c_stride", "out_l_stride")]
        + [("conv_state", _vp)]
        + [(n, _i64) for n in ("conv_state_batch_stride", "conv_state_c_stride", "conv_state_l_stride")]
        + [("reverse_from", _i32), ("reserved1", _i32)]
    )


class ConvBwdParams(ctypes.Structure):
    _fields_ = (
        [("f", ConvFwdParams), ("dout", _vp), ("dx", _vp), ("dweight", _fp), ("dbias", _fp)]
        + [(n, _i64) for n in ("dout_batch_stride", "dout_c_stride", "dout_l_stride",
                               "dx_batch_stride", "dx_c_stride", "dx_l_stride",
                               "dweight_c_stride", "dweight_width_stride")]
        + [("dx_accumulate", _i32), ("reserved0", _i32)]
    )


class ConvFwdDualParams(ctypes.Structure):
    _fields_ = ([("f", ConvFwdParams)] + [(n, _vp) for n in ("weight_b", "bias_b", "out_b")]
                + [(n, _i64) for n in ("weight_b_c_stride", "weight_b_width_stride", "out_b_batch_stride", "out_b_c_stride")])


class ConvXprojDualParams(ctypes.Structure):
    _fields_ = ([("c", ConvFwdDualParams)] + [(n, _vp) for n in ("w_x", "w_x_b", "x_dbl", "x_dbl_b")]
                + [(n, _i32) for n in ("m", "tile")] + [(n, _i64) for n in ("wx_row_stride", "xdbl_batch_stride", "xdbl_row_stride")])


class NormParams(ctypes.Structure):
    _fields_ = (
        [(n, _i32) for n in ("rows", "cols", "x_dtype", "res_dtype", "is_rms")] + [("eps", ctypes.c_float)]
        + [(n, _vp) for n in ("x", "residual", "weight", "bias", "y", "residual_out", "mean", "rstd")]
        + [(n, _i64) for n in ("x_row_stride", "residual_row_stride", "y_row_stride", "residual_out_row_stride")]
    )


class NormBwdParams(ctypes.Structure):
    _fields_ = (
        [("f", NormParams)]
        + [(n, _vp) for n in ("s", "dy", "dres_out", "dx", "dres_in", "dw_partial", "db_partial")]
        + [("n_partials", _i32), ("reserved", _i32)]
        + [(n, _i64) for n in ("s_row_stride", "dy_row_stride", "dres_out_row_stride", "dx_row_stride",
                               "dres_in_row_stride")]
    )


class StateUpdateParams(ctypes.Structure):
    _fields_ = (
        [(n, _i32) for n in ("batch", "dim", "dstate", "state_dtype", "x_dtype", "bc_dtype", "w_dtype", "dt_softplus",
                             "dt_dtype", "z_dtype")]
        + [(n, _vp) for n in ("state", "x", "dt", "A", "B", "C", "D", "z", "dt_bias", "out")]
        + [(n, _i64) for n in ("state_batch_stride", "state_d_stride", "state_n_stride", "x_batch_stride",
                               "x_d_stride", "dt_batch_stride", "dt_d_stride", "z_batch_stride", "z_d_stride",
                               "out_batch_stride", "out_d_stride", "A_d_stride", "A_n_stride", "B_batch_stride",
                               "B_n_stride", "C_batch_stride", "C_n_stride")]
    )


class ProjApplyParams(ctypes.Structure):
    _fields_ = (
        [(n, _i32) for n in ("batch", "rows", "k", "seqlen", "dtype", "accumulate", "tiles_per_wg", "reserved")]
        + [(n, _vp) for n in ("w", "inp", "out")]
        + [(n, _i64) for n in ("w_row_stride", "w_k_stride", "in_batch_stride", "in_k_stride", "out_batch_stride",
                               "out_row_stride")]
    )


class ProjWgradParams(ctypes.Structure):
    _fields_ = (
        [(n, _i32) for n in ("batch", "m", "n", "seqlen", "dtype", "tiles_per_wg", "dw_transposed", "reserved")]
        + [(n, _vp) for n in ("p", "q", "dw")]
        + [(n, _i64) for n in ("p_batch_stride", "p_row_stride", "q_batch_stride", "q_row_stride", "dw_row_stride")]
    )


class ProjKredParams(ctypes.Structure):
    _fields_ = (
        [(n, _i32) for n in ("batch", "m", "k", "seqlen", "dtype", "tile")]
        + [(n, _vp) for n in ("w", "inp", "out", "w2", "inp2", "out2")]
        + [(n, _i64) for n in ("w_row_stride", "w_k_stride", "in_batch_stride", "in_k_stride", "out_batch_stride",
                               "out_row_stride")]
        + [(n, _vp) for n in ("cast_src", "cast_src2")]
        + [(n, _i32) for n in ("cast_rows", "cast_groups")]
        + [(n, _i64) for n in ("cast_group_stride", "cast_batch_stride", "cast_row_stride")]
    )


class ProjConvBwdParams(ctypes.Structure):
    _fields_ = (
        [(n, _i32) for n in ("batch", "dim", "k", "seqlen", "width", "dtype", "wdtype", "reverse", "reverse_from",
                             "dx_accumulate", "tiles_per_wg", "reserved")]
        + [(n, _vp) for n in ("x", "du", "dx_dbl", "w_x", "conv_weight", "conv_bias", "dx", "dconv_weight", "dconv_bias", "dw_x")]
        + [(n, _i64) for n in ("x_batch_stride", "x_c_stride", "du_batch_stride", "du_c_stride", "dxdbl_batch_stride",
                               "dxdbl_k_stride", "wx_k_stride", "wx_c_stride", "conv_weight_c_stride",
                               "conv_weight_width_stride", "dx_batch_stride", "dx_c_stride", "dconv_weight_c_stride",
                               "dconv_weight_width_stride", "dwx_k_stride")]
    )


class PrepJob(ctypes.Structure):
    _fields_ = [("src", _vp), ("dst", _vp), ("rows", _i32), ("cols", _i32), ("src_row_stride", _i64), ("dst_row_stride", _i64),
                ("src_dtype", _i32), ("dst_dtype", _i32), ("op", _i32), ("dst_col_stride", _i32)]


PREP_MAX_JOBS = 8
PREP_CAST, PREP_CAST_T, PREP_NEG_EXP = 0, 1, 2


class PrepParams(ctypes.Structure):
    _fields_ = [("n_jobs", _i32), ("reserved", _i32), ("job", PrepJob * PREP_MAX_JOBS)]


EXPORTS = (
    "vms_selective_scan_fwd", "vms_selective_scan_bwd", "vms_causal_conv1d_fwd", "vms_causal_conv1d_bwd",
    "vms_causal_conv1d_update", "vms_abi_version", "vms_last_error", "vms_sizeof_scan_fwd_params",
    "vms_sizeof_scan_bwd_params", "vms_sizeof_conv_fwd_params", "vms_sizeof_conv_bwd_params",
    "vms_scan_fwd_workspace_bytes", "vms_scan_bwd_workspace_bytes", "vms_scan_x_elems", "vms_scan_x_pitch",
    "vms_layer_norm_fwd", "vms_layer_norm_bwd", "vms_layer_norm_bwd_partials", "vms_sizeof_norm_params",
    "vms_sizeof_norm_bwd_params", "vms_selective_state_update", "vms_sizeof_state_update_params",
    "vms_last_kernel", "vms_build_flags",
    "vms_proj_apply", "vms_proj_wgrad", "vms_sizeof_proj_apply_params", "vms_sizeof_proj_wgrad_params",
    "vms_proj_conv_bwd", "vms_sizeof_proj_conv_bwd_params",
    "vms_param_prep", "vms_sizeof_prep_params",
    "vms_causal_conv1d_fwd_dual", "vms_sizeof_conv_fwd_dual_params",
    "vms_selective_scan_bwd_dual", "vms_scan_bwd_dual_fused",
    "vms_proj_kred", "vms_sizeof_proj_kred_params",
    "vms_conv_xproj_dual", "vms_sizeof_conv_xproj_dual_params", "vms_layer_norm_bwd_finish", "vms_sum_slices",
)

ABI_VERSION = 11   # include/vms_hip.h VMS_ABI_VERSION: checked against libvms_hip.so and against the compiled binding
IMPL_AUTO, IMPL_GENERIC, IMPL_PAIR = 0, 1, 3
_IMPL_NAMES = {"g": IMPL_GENERIC, "p": IMPL_PAIR}


class _Debug:
    """Every test / profiling switch of the Python layers, in one place (`vms_hip.debug`).  Tests set attributes with
    `monkeypatch.setattr(vms_hip.debug, name, value)`; A/B runs of the tools pass `VMS_DEBUG="name=value,name=value"`, parsed
    ONCE at import.  The C ABI reads no environment at all (kernel generation and range counts travel in the parameter block).
    User-facing environment variables are only VMS_HIP_LIB (another build of the library), VMS_X_LAYOUT (checkpoint layout policy:
    1 | 3 | auto) and VMS_CHECKPOINT_LVL (the modules' recompute policy).

    scan_impl        None | "generic" | "pair": vms_scan_impl of every scan call (vms_hip.h; None = the library's choice)
    force_generic    every scan on the generic kernels (the reference's full contract)
    fwd_segments     forced range count of the sequence-split forward scan (0 = the library's choice, 1 = never split)
    bwd_segments     the same for the backward scan
    no_torch_ext     serve every call through the ctypes binding instead of the compiled one (_vms_torch.so)
    no_inner_ext     the fused inner nodes as Python compositions instead of the compiled one-call node
    mfma_proj        True / False: the inner node's dt_proj products on the hand-written MFMA kernels / the library; None = by shape
    no_fused_tail    the backward's tail as three kernels (library GEMMs + conv1d backward) instead of vms_proj_conv_bwd
    no_proj_kred     x_proj / d_dt as library GEMMs instead of vms_proj_kred
    no_dual_bwd      one backward-scan launch per direction of a bidirectional block
    no_dual_conv     one conv1d launch per direction
    no_conv_xproj    conv1d and x_proj of a bidirectional block as separate launches
    no_reverse       modules: the backward direction on flipped copies (the reference's way) instead of the kernels' reverse mode
    dbm_two_nodes    the DBM block as one node per direction instead of the stacked one-node form
    no_param_prep    every node prepares its own parameters instead of one vms_param_prep launch per block
    no_seq_pad       ragged sequences as they come instead of padded to whole vectors inside the mixer
    no_seq_pad_tiles pad to the next multiple of 16 only (not up to whole 256-token GEMM tiles)

    WHEN a switch is read: scan_impl, force_generic, *_segments, mfma_proj, no_fused_tail, no_inner_ext and no_torch_ext are
    read at every call -- setting the attribute takes effect at once.  The others are IMPORT-TIME switches: the layers copy them into
    module constants when they are first imported (selective_scan_interface._DUAL_BWD / _DUAL_CONV / _CONV_XPROJ / _PROJ_KRED,
    modules._core._USE_REVERSE_KERNELS / _DBM_STACKED / _PARAM_PREP / _SEQ_PAD / _SEQ_PAD_TILES), so only VMS_DEBUG (parsed before
    those imports) or a monkeypatch of the module constant itself changes them -- which is what the tests do.
    """
    _SWITCHES = ("scan_impl", "force_generic", "fwd_segments", "bwd_segments", "no_torch_ext", "no_inner_ext", "mfma_proj", "no_fused_tail",
                 "no_proj_kred", "no_dual_bwd", "no_dual_conv", "no_conv_xproj", "no_reverse", "dbm_two_nodes", "no_param_prep", "no_seq_pad",
                 "no_seq_pad_tiles")
    scan_impl = None
    force_generic = False
    fwd_segments = 0
    bwd_segments = 0
    no_torch_ext = False
    no_inner_ext = False
    mfma_proj = None
    no_fused_tail = False
    no_proj_kred = False
    no_dual_bwd = False
    no_dual_conv = False
    no_conv_xproj = False
    no_reverse = False
    dbm_two_nodes = False
    no_param_prep = False
    no_seq_pad = False
    no_seq_pad_tiles = False

    def __init__(self, spec=""):
        for item in filter(None, (t.strip() for t in spec.split(","))):
            k, _, v = item.partition("=")
            if k not in self._SWITCHES:   # (an explicit list: hasattr() also accepted __doc__, __init__, ...)
                raise ValueError(f"VMS_DEBUG: unknown switch {k!r} (see vms_hip.debug.__doc__)")
            cur = getattr(type(self), k)
            if k == "scan_impl":
                val = v or None
            elif k == "mfma_proj":
                val = None if v == "" else v not in ("0", "false", "False")
            elif isinstance(cur, bool):
                val = v not in ("0", "false", "False")
            else:
                val = int(v)
            setattr(self, k, val)


debug = _Debug(os.environ.get("VMS_DEBUG", ""))


def scan_impl_from_env():
    """vms_scan_impl of the next scan call (name kept from the time this read the environment per call: see `debug`)"""
    if debug.force_generic:
        return IMPL_GENERIC
    e = debug.scan_impl
    return IMPL_AUTO if not e else _IMPL_NAMES.get(e[0], IMPL_AUTO)


def _segments_from_env(name):
    n = debug.fwd_segments if "FWD" in name else debug.bwd_segments
    return max(int(n), 1) if n else 0


def last_kernel():
    """Kernel family enqueued by this thread's last successful launch call (vms_hip.h vms_last_kernel)."""
    return lib().vms_last_kernel().decode()


def segments_from_env(name):
    return _segments_from_env(name)

_lib = None
_ext = False   # the compiled PyTorch binding (_vms_torch.so, csrc/torch_binding): False = not looked for yet, None = absent


def ext():
    """The compiled binding of the scan / conv entry points, or None when it has not been built (or debug.no_torch_ext):
    the ctypes path below then serves every call.  Both end in the same C ABI."""
    global _ext
    if _ext is False:
        _ext = None
        if not debug.no_torch_ext and LIB_PATH == _DEFAULT_LIB:   # A/B libraries (VMS_HIP_LIB): ctypes only
            lib()
            try:
                import _vms_torch
                if _vms_torch.abi_version() == ABI_VERSION:
                    _ext = _vms_torch
                else:   # a stale build must not go unnoticed either
                    import warnings
                    warnings.warn(f"_vms_torch.so speaks ABI {_vms_torch.abi_version()}, this package {ABI_VERSION}: rebuild it "
                                  "with csrc/torch_binding/build.py --force.  Falling back to the ctypes binding.")
            except ImportError as e:
                # absent: fine (ctypes serves the calls).  Present but unloadable -- built against another torch / Python,
                # see csrc/torch_binding/build.py's stamp -- must not go unnoticed: every launch pays ~40 us more.
                so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "_vms_torch.so")
                if os.path.exists(so):
                    import warnings
                    warnings.warn(f"_vms_torch.so exists but does not import ({e}); rebuild it with "
                                  "csrc/torch_binding/build.py --force.  Falling back to the ctypes binding.")
    return _ext


# Optional per-call device timing (used by bench.py): when a list is installed here every C-ABI
# call is bracketed by two events on the stream the kernel is launched on.
_timing = None
_timing_only = None


def start_timing(reserve=0, only=None):
    """reserve: launches expected until stop_timing() -- the compiled binding creates that many event pairs now, outside
    the timed region (it grows the pool on demand if more arrive).  only: time this entry point alone (e.g.
    "vms_selective_scan_bwd"): every timed launch puts two event records into the stream, ~6 us of idle GPU around it."""
    global _timing, _timing_only
    _timing, _timing_only = [], only
    if ext() is not None:
        ext().timing_start(int(reserve), only or "")


def stop_timing():
    """-> {entry point: [milliseconds per call]}; synchronises the device."""
    global _timing
    rec, _timing = _timing, None
    torch.cuda.synchronize()
    out = {}
    for name, e0, e1 in rec or []:
        out.setdefault(name, []).append(e0.elapsed_time(e1))
    if ext() is not None:
        for name, ms in ext().timing_stop():
            out.setdefault(name, []).append(ms)
    return out


def lib():
    """Load libvms_hip.so once; fail loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build the HIP library first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C video-mamba-suite_amd/csrc)")
        L = ctypes.CDLL(LIB_PATH)
        L.vms_last_error.restype = ctypes.c_char_p
        L.vms_last_kernel.restype = ctypes.c_char_p
        if L.vms_abi_version() != ABI_VERSION:
            raise ImportError(f"{LIB_PATH} has ABI version {L.vms_abi_version()}, this binding speaks {ABI_VERSION}: rebuild it")
        for name, st in (("scan_fwd", ScanFwdParams), ("scan_bwd", ScanBwdParams),
                         ("conv_fwd", ConvFwdParams), ("conv_bwd", ConvBwdParams),
                         ("norm", NormParams), ("norm_bwd", NormBwdParams), ("state_update", StateUpdateParams),
                         ("proj_apply", ProjApplyParams), ("proj_wgrad", ProjWgradParams),
                         ("proj_conv_bwd", ProjConvBwdParams), ("prep", PrepParams), ("conv_fwd_dual", ConvFwdDualParams),
                         ("proj_kred", ProjKredParams), ("conv_xproj_dual", ConvXprojDualParams)):
            n = getattr(L, f"vms_sizeof_{name}_params")()
            if n != ctypes.sizeof(st):
                raise ImportError(f"ABI mismatch: {name} params are {n} bytes in the library, "
                                  f"{ctypes.sizeof(st)} in the binding")
        for fn in EXPORTS[:5]:
            getattr(L, fn).restype = ctypes.c_int
        for fn in EXPORTS[11:14]:
            getattr(L, fn).restype = ctypes.c_int64
        L.vms_scan_x_pitch.restype = ctypes.c_int64
        L.vms_scan_x_pitch.argtypes = [ctypes.c_void_p, ctypes.c_int32]
        _lib = L
    return _lib


def _call(fn_name, params, ref_tensor, params2=None):
    """params2: the second parameter block of the two-block entry points (vms_selective_scan_bwd_dual)"""
    if params2 is not None:
        return _call2(fn_name, params, params2, ref_tensor)
    L = lib()
    if not ref_tensor.is_cuda:
        raise RuntimeError(f"{fn_name}: tensors must be on a GPU (no CPU path in this library)")
    idx = ref_tensor.device.index
    if idx == torch.cuda.current_device():
        # the common case without the device guard and stream object: ~20 us less host time per call, which is
        # what small problems are bound by.  Timing (bench.py) only adds two event records on the same raw stream.
        if _timing is None or (_timing_only and _timing_only != fn_name):
            rc = getattr(L, fn_name)(ctypes.byref(params), ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(idx)))
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = getattr(L, fn_name)(ctypes.byref(params), ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(idx)))
            e1.record()
            _timing.append((fn_name, e0, e1))
    else:
        with torch.cuda.device(ref_tensor.device):
            cur = torch.cuda.current_stream()
            timed = _timing is not None and not (_timing_only and _timing_only != fn_name)
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)
            rc = getattr(L, fn_name)(ctypes.byref(params), ctypes.c_void_p(cur.cuda_stream))
            if timed:
                e1.record(cur)
                _timing.append((fn_name, e0, e1))
    if rc != 0:
        raise RuntimeError(f"{fn_name} failed (status {rc}): {L.vms_last_error().decode()}")


def _call_plain(fn_name, ref_tensor, *args):
    """the entry points with a plain argument list (no parameter block): args, then the stream of ref_tensor's device"""
    L = lib()
    idx = ref_tensor.device.index
    if idx == torch.cuda.current_device():
        rc = getattr(L, fn_name)(*args, ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(idx)))
    else:
        with torch.cuda.device(ref_tensor.device):
            rc = getattr(L, fn_name)(*args, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0:
        raise RuntimeError(f"{fn_name} failed (status {rc}): {L.vms_last_error().decode()}")


def _call2(fn_name, pa, pb, ref_tensor):
    L = lib()
    if not ref_tensor.is_cuda:
        raise RuntimeError(f"{fn_name}: tensors must be on a GPU (no CPU path in this library)")
    with torch.cuda.device(ref_tensor.device):
        cur = torch.cuda.current_stream()
        timed = _timing is not None and not (_timing_only and _timing_only != fn_name)
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(cur)
        rc = getattr(L, fn_name)(ctypes.byref(pa), ctypes.byref(pb), ctypes.c_void_p(cur.cuda_stream))
        if timed:
            e1.record(cur)
            _timing.append((fn_name, e0, e1))
    if rc != 0:
        raise RuntimeError(f"{fn_name} failed (status {rc}): {L.vms_last_error().decode()}")


def _ptr(t):
    return None if t is None else t.data_ptr()


def dtype_code(t):
    try:
        return _DTYPE[t.dtype]
    except KeyError:
        raise RuntimeError(f"unsupported dtype {t.dtype}: expected float32, float16 or bfloat16") from None


# ---- which checkpoints a forward leaves for its backward (vms_hip.h x_has_sub) ----------------------------------------------
# "fine"   = the state after every 8 elements (x_has_sub == 3): 8 * batch * dim * seqlen BYTES per scan at d_state 16 -- as much
#            again as the scan's own saved activations (537 MB per direction at (8, 1024, 8192), the reference's x is 4 MB) --
#            for a backward scan ~10 % faster;
# "coarse" = the state after every 128 elements (x_has_sub == 1, 1/16 of that);
# "auto"   (default) = fine while the device has room to spare, coarse otherwise: fine only if, on the tensors' device, no more
#            than a quarter of the memory is allocated and this scan's fine checkpoints take under 1/8 of what is unallocated.
#            A stack that fits with the coarse layout in under ~7/8 of the device's memory therefore still fits; training runs
#            sized closer to the limit than that should pin "coarse".
# VMS_X_LAYOUT = 1 | 3 (environment) overrides everything; set_x_layout_policy() sets the process default; the
# x_layout_policy(...) context (what the Mamba modules' `scan_checkpoints=` argument uses) overrides it per thread.
import threading

_x_policy = "auto"
_x_tls = threading.local()
_total_mem = {}
_POLICIES = {"auto": "auto", "fine": "fine", "3": "fine", "coarse": "coarse", "1": "coarse"}


def set_x_layout_policy(policy):
    global _x_policy
    _x_policy = _POLICIES[str(policy)]


class x_layout_policy:
    def __init__(self, policy):
        self.policy = None if policy is None else _POLICIES[str(policy)]

    def __enter__(self):
        self.prev = getattr(_x_tls, "policy", None)
        if self.policy is not None:
            _x_tls.policy = self.policy

    def __exit__(self, *exc):
        _x_tls.policy = self.prev


def current_x_layout_policy():
    env = os.environ.get("VMS_X_LAYOUT")
    if env in _POLICIES:
        return _POLICIES[env]
    return getattr(_x_tls, "policy", None) or _x_policy


def x_mode_for_shape(batch, dim, seqlen, dstate, device, for_backward=True):
    """-> the `mode` argument of vms_scan_x_pitch: 1 = the 128-element checkpoints, -1 = the 8-element ones where the backward
    kernel that reads them takes the problem (the library decides that part)."""
    if not for_backward:
        return 1
    pol = current_x_layout_policy()
    if pol != "auto":
        return 1 if pol == "coarse" else -1
    dev = torch.device(device)
    if dev.type != "cuda":
        return -1
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    total = _total_mem.get(idx)
    if total is None:
        total = _total_mem[idx] = torch.cuda.get_device_properties(idx).total_memory
    # what the 8-element layout allocates: 258 * dstate floats per (row, 2048-element chunk) whatever the row's length (a short
    # row pays for a whole chunk: 16.5 KB at dstate 16)
    need = batch * dim * ((seqlen + 2047) // 2048) * 258 * dstate * 4
    try:
        used = _allocated_bytes(idx)
    except RuntimeError:
        used = None
    if used is None:         # nothing known about the device's memory: the layout that cannot surprise
        return 1
    return -1 if (4 * used <= total and 8 * need <= total - used) else 1


def _allocated_bytes(idx):
    """torch.cuda.memory_allocated without its flattening of the allocator's whole statistics tree into a dict (100 us per call,
    twice per block step: a fifth of a block's host time)"""
    try:
        return torch._C._cuda_memoryStats(idx)["allocated_bytes"]["all"]["current"]
    except (AttributeError, KeyError, TypeError):
        pass
    try:
        return torch.cuda.memory_allocated(idx)
    except RuntimeError:     # a pluggable allocator (torch.cuda.memory.CUDAPluggableAllocator) keeps no statistics
        return None


def x_mode_for(u, dstate, for_backward=True):
    return x_mode_for_shape(u.shape[0], u.shape[1], u.shape[2], dstate, u.device, for_backward)


def x_mode_from_env():
    """(kept for tools) the mode the policy gives without a size: VMS_X_LAYOUT=1 -> 1, else the library's choice"""
    return 1 if current_x_layout_policy() == "coarse" else -1


def x_layout_of(x, dstate):
    """vms_hip.h x_has_sub of a (batch, dim, n_chunks, 2 * dstate) view: what its pitch has room for."""
    if x.stride(3) != 1:
        return 0
    if x.is_complex():   # vms_hip.h is_complex: the state after every 512 elements behind the reference-shaped slots
        return 1 if x.stride(2) >= 6 * dstate else 0
    return 3 if x.stride(2) >= 258 * dstate else (1 if x.stride(2) >= 18 * dstate else 0)


def fill_scan_fwd(P, u, delta, A, B, C, D, z, delta_bias, out, out_z, x, delta_softplus, reverse=False, reverse_from=0):
    batch, dim, seqlen = u.shape
    dstate = A.shape[1]
    var_B, var_C = B.dim() >= 3, C.dim() >= 3
    P.batch, P.dim, P.seqlen, P.dstate = batch, dim, seqlen, dstate
    P.n_groups = B.shape[1] if var_B else (C.shape[1] if var_C else 1)
    P.n_chunks = (seqlen + 2047) // 2048
    P.dtype = dtype_code(u)
    P.is_variable_B, P.is_variable_C, P.delta_softplus = int(var_B), int(var_C), int(bool(delta_softplus))
    P.reverse = int(bool(reverse))
    P.reverse_from = int(reverse_from)
    P.is_complex = int(A.is_complex())
    P.impl = scan_impl_from_env()
    P.u, P.delta, P.A, P.B, P.C = _ptr(u), _ptr(delta), _ptr(A), _ptr(B), _ptr(C)
    P.D, P.z, P.delta_bias = _ptr(D), _ptr(z), _ptr(delta_bias)
    P.out, P.out_z, P.x = _ptr(out), _ptr(out_z), _ptr(x)
    if x is not None:
        # x is dense (.., 2N), or the (.., 2N) view of a (.., 18N) buffer carrying 128-element
        # sub-checkpoints behind the reference-shaped slots, or the dense tensor followed by the
        # row-major kernels' checkpoint region (see include/vms_hip.h)
        P.x_chunk_stride = x.stride(2)
        P.x_has_sub = x_layout_of(x, dstate)
    P.u_batch_stride, P.u_d_stride = u.stride(0), u.stride(1)
    P.delta_batch_stride, P.delta_d_stride = delta.stride(0), delta.stride(1)
    if z is not None:
        P.z_batch_stride, P.z_d_stride = z.stride(0), z.stride(1)
    if out is not None:
        P.out_batch_stride, P.out_d_stride = out.stride(0), out.stride(1)
    if out_z is not None:
        P.out_z_batch_stride, P.out_z_d_stride = out_z.stride(0), out_z.stride(1)
    P.A_d_stride, P.A_dstate_stride = A.stride(0), A.stride(1)
    if var_B:
        P.B_batch_stride, P.B_group_stride, P.B_dstate_stride = B.stride(0), B.stride(1), B.stride(2)
    else:
        P.B_d_stride, P.B_dstate_stride = B.stride(0), B.stride(1)
    if var_C:
        P.C_batch_stride, P.C_group_stride, P.C_dstate_stride = C.stride(0), C.stride(1), C.stride(2)
    else:
        P.C_d_stride, P.C_dstate_stride = C.stride(0), C.stride(1)




def _ws_bytes(fn_name, params, ref_tensor):
    """Workspace query ON THE TENSORS' DEVICE: the split decision behind it reads the current device's CU count."""
    fn = getattr(lib(), fn_name)
    if not ref_tensor.is_cuda or ref_tensor.device.index == torch.cuda.current_device():
        return fn(ctypes.byref(params))
    with torch.cuda.device(ref_tensor.device):
        return fn(ctypes.byref(params))


def scan_fwd(u, delta, A, B, C, D, z, delta_bias, out, out_z, x, delta_softplus, reverse=False,
             out_z_accumulate=False, bc_pad=0, reverse_from=0, for_backward=True):
    """x is None: this function chooses the checkpoint layout, allocates x and returns it (for_backward = False: the
    small layout -- nothing will read the 8-element checkpoints)."""
    P = ScanFwdParams()
    fill_scan_fwd(P, u, delta, A, B, C, D, z, delta_bias, out, out_z, x, delta_softplus, reverse, reverse_from)
    P.out_z_accumulate = int(bool(out_z_accumulate))
    P.bc_pad = int(bc_pad)
    P.segments = _segments_from_env("VMS_FWD_SEGMENTS")
    ws = None
    if x is None:
        batch, dim, n_chunks, dstate = P.batch, P.dim, P.n_chunks, P.dstate
        # the reference-shaped tensor is a view of a wider buffer whose tail carries the checkpoints for the backward
        # kernel (include/vms_hip.h)
        pitch = lib().vms_scan_x_pitch(ctypes.byref(P), x_mode_for(u, dstate, for_backward))
        x = torch.empty(batch, dim, n_chunks, pitch, device=u.device, dtype=torch.float32)[..., :dstate * 2]
        P.x, P.x_chunk_stride = _ptr(x), x.stride(2)
        P.x_has_sub = x_layout_of(x, dstate)
    if not P.workspace:
        nws = _ws_bytes("vms_scan_fwd_workspace_bytes", P, u)   # state carries of a sequence-split forward
        if nws > 0:
            ws = torch.empty(nws, device=u.device, dtype=torch.uint8)
            P.workspace, P.workspace_bytes = _ptr(ws), nws
    _call("vms_selective_scan_fwd", P, u)
    return x


def _fill_scan_bwd(u, delta, A, B, C, D, z, delta_bias, dout, x, out, out_z, du, ddelta, dA, dB, dC, dD,
                   ddelta_bias, dz, delta_softplus, reverse=False, dz_accumulate=False, bc_pad=0, reverse_from=0):
    Q = ScanBwdParams()
    fill_scan_fwd(Q.f, u, delta, A, B, C, D, z, delta_bias, out, out_z, x, delta_softplus, reverse, reverse_from)
    Q.dout, Q.du, Q.ddelta, Q.dz = _ptr(dout), _ptr(du), _ptr(ddelta), _ptr(dz)
    Q.dA, Q.dB, Q.dC, Q.dD, Q.ddelta_bias = _ptr(dA), _ptr(dB), _ptr(dC), _ptr(dD), _ptr(ddelta_bias)
    Q.dout_batch_stride, Q.dout_d_stride = dout.stride(0), dout.stride(1)
    Q.du_batch_stride, Q.du_d_stride = du.stride(0), du.stride(1)
    Q.ddelta_batch_stride, Q.ddelta_d_stride = ddelta.stride(0), ddelta.stride(1)
    if dz is not None:
        Q.dz_batch_stride, Q.dz_d_stride = dz.stride(0), dz.stride(1)
    Q.dA_d_stride, Q.dA_dstate_stride = dA.stride(0), dA.stride(1)
    if B.dim() >= 3:
        Q.dB_batch_stride, Q.dB_group_stride, Q.dB_dstate_stride = dB.stride(0), dB.stride(1), dB.stride(2)
    else:
        Q.dB_d_stride, Q.dB_dstate_stride = dB.stride(0), dB.stride(1)
    if C.dim() >= 3:
        Q.dC_batch_stride, Q.dC_group_stride, Q.dC_dstate_stride = dC.stride(0), dC.stride(1), dC.stride(2)
    else:
        Q.dC_d_stride, Q.dC_dstate_stride = dC.stride(0), dC.stride(1)
    Q.dz_accumulate = int(bool(dz_accumulate))
    Q.f.bc_pad = int(bc_pad)
    Q.f.segments = _segments_from_env("VMS_BWD_SEGMENTS")
    return Q


def _scan_bwd_workspace(Q, u):
    nws = _ws_bytes("vms_scan_bwd_workspace_bytes", Q, u)   # adjoint carries of a sequence-split backward
    ws = None
    if nws > 0:
        ws = torch.empty(nws, device=u.device, dtype=torch.uint8)
        Q.f.workspace, Q.f.workspace_bytes = _ptr(ws), nws
    return ws


def scan_bwd(u, delta, A, B, C, D, z, delta_bias, dout, x, out, out_z, du, ddelta, dA, dB, dC, dD,
             ddelta_bias, dz, delta_softplus, reverse=False, dz_accumulate=False, bc_pad=0, reverse_from=0):
    Q = _fill_scan_bwd(u, delta, A, B, C, D, z, delta_bias, dout, x, out, out_z, du, ddelta, dA, dB, dC, dD,
                       ddelta_bias, dz, delta_softplus, reverse, dz_accumulate, bc_pad, reverse_from)
    ws = _scan_bwd_workspace(Q, u)   # noqa: F841 -- alive until the launch is enqueued
    _call("vms_selective_scan_bwd", Q, u)


def scan_bwd_dual(args_a, args_b):
    """vms_selective_scan_bwd_dual: args_a / args_b = the positional arguments of scan_bwd for the left-to-right and the
    right-to-left direction of one bidirectional block (b's dz: None).  -> True when the pair ran as ONE grid."""
    Qa, Qb = _fill_scan_bwd(*args_a), _fill_scan_bwd(*args_b)
    u = args_a[0]
    with torch.cuda.device(u.device):
        fused = bool(lib().vms_scan_bwd_dual_fused(ctypes.byref(Qa), ctypes.byref(Qb)))
    ws = None if fused else (_scan_bwd_workspace(Qa, u), _scan_bwd_workspace(Qb, u))   # noqa: F841
    _call("vms_selective_scan_bwd_dual", Qa, u, Qb)
    return fused


def fill_conv_fwd(P, x, weight, bias, out, silu, reverse=False, reverse_from=0):
    P.batch, P.dim, P.seqlen = x.shape
    P.width = weight.shape[-1]
    P.dtype, P.wdtype = dtype_code(x), dtype_code(weight)
    P.silu_activation = int(bool(silu))
    P.reverse = int(bool(reverse))
    P.reverse_from = int(reverse_from)
    P.x, P.weight, P.bias, P.out = _ptr(x), _ptr(weight), _ptr(bias), _ptr(out)
    P.x_batch_stride, P.x_c_stride, P.x_l_stride = x.stride()
    P.weight_c_stride, P.weight_width_stride = weight.stride()
    if out is not None:
        P.out_batch_stride, P.out_c_stride, P.out_l_stride = out.stride()


def conv_fwd(x, weight, bias, out, silu, reverse=False, reverse_from=0):
    P = ConvFwdParams()
    fill_conv_fwd(P, x, weight, bias, out, silu, reverse, reverse_from)
    _call("vms_causal_conv1d_fwd", P, x)


def conv_fwd_dual(x, weight, bias, out, weight_b, bias_b, out_b, silu):
    """out = the causal filter (weight, bias), out_b = the anti-causal one (weight_b, bias_b) of the same x, one pass (vms_hip.h)"""
    Q = ConvFwdDualParams()
    fill_conv_fwd(Q.f, x, weight, bias, out, silu)
    if weight_b.dtype != weight.dtype or weight_b.shape != weight.shape or (bias is None) != (bias_b is None):
        raise RuntimeError("conv_fwd_dual: the two filters need the same dtype, shape and bias presence")
    Q.weight_b, Q.bias_b, Q.out_b = _ptr(weight_b), _ptr(bias_b), _ptr(out_b)
    Q.weight_b_c_stride, Q.weight_b_width_stride = weight_b.stride()
    if out_b.stride(2) != 1 or out_b.shape != x.shape:
        raise RuntimeError("conv_fwd_dual: out_b must be (batch, dim, seqlen) with a unit seqlen stride")
    Q.out_b_batch_stride, Q.out_b_c_stride = out_b.stride(0), out_b.stride(1)
    _call("vms_causal_conv1d_fwd_dual", Q, x)


def conv_xproj_dual_eligible(x, weight, bias, weight_b, bias_b, w_x, w_x_b):
    """Does vms_conv_xproj_dual take the head of a bidirectional block's forward?  (16-bit x of one dtype with the x_proj weights, conv
    weights in fp32 or that dtype, unit seqlen stride, seqlen / dim / strides multiples of 8, 16-byte aligned, m <= 96)"""
    if not (x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and x.dim() == 3 and x.stride(2) == 1 and w_x.dtype == x.dtype
            and w_x_b.dtype == x.dtype and weight.dtype in (torch.float32, x.dtype) and weight_b.dtype == weight.dtype
            and weight.shape == weight_b.shape and weight.dim() == 2 and 2 <= weight.shape[1] <= 4 and (bias is None) == (bias_b is None)):
        return False
    if (bias is not None and (bias.dtype != weight.dtype or bias_b.dtype != weight.dtype)) or w_x.shape != w_x_b.shape or w_x.stride() != w_x_b.stride():
        return False
    b, d, L = x.shape
    m = w_x.shape[0]
    if not (w_x.dim() == 2 and w_x.shape[1] == d and 1 <= m <= 96 and w_x.stride(1) == 1 and w_x.stride(0) % 8 == 0 and weight.shape[0] == d):
        return False
    if L % 8 or d % 8 or x.stride(0) % 8 or x.stride(1) % 8 or x.stride(1) < L or x.data_ptr() % 16 or w_x.data_ptr() % 16 or w_x_b.data_ptr() % 16:
        return False
    return ((d - 1) * x.stride(1) + L) * 2 < 2 ** 31 and ((d - 1) * L + L) * 2 < 2 ** 31


def conv_xproj_dual(x, weight, bias, out, weight_b, bias_b, out_b, w_x, w_x_b, x_dbl, x_dbl_b, tile=0):
    """out / out_b = both directions' causal_conv1d (+ SiLU) of x, x_dbl / x_dbl_b = w_x @ out / w_x_b @ out_b, one pass over x
    (vms_hip.h vms_conv_xproj_dual)."""
    if not conv_xproj_dual_eligible(x, weight, bias, weight_b, bias_b, w_x, w_x_b):
        raise RuntimeError("conv_xproj_dual: not a problem this entry point takes (vms_hip.conv_xproj_dual_eligible)")
    Q = ConvXprojDualParams()
    fill_conv_fwd(Q.c.f, x, weight, bias, out, True)
    Q.c.weight_b, Q.c.bias_b, Q.c.out_b = _ptr(weight_b), _ptr(bias_b), _ptr(out_b)
    Q.c.weight_b_c_stride, Q.c.weight_b_width_stride = weight_b.stride()
    if out.stride(2) != 1 or out_b.stride(2) != 1 or out.shape != x.shape or out_b.shape != x.shape or out.dtype != x.dtype or out_b.dtype != x.dtype:
        raise RuntimeError("conv_xproj_dual: out / out_b must be (batch, dim, seqlen) in x's dtype with a unit seqlen stride")
    Q.c.out_b_batch_stride, Q.c.out_b_c_stride = out_b.stride(0), out_b.stride(1)
    m = w_x.shape[0]
    for t in (x_dbl, x_dbl_b):
        if tuple(t.shape) != (x.shape[0], m, x.shape[2]) or t.dtype != x.dtype or t.stride(2) != 1 or t.stride() != x_dbl.stride():
            raise RuntimeError("conv_xproj_dual: x_dbl / x_dbl_b must be (batch, m, seqlen) in x's dtype with equal strides, unit seqlen stride")
    Q.w_x, Q.w_x_b, Q.x_dbl, Q.x_dbl_b = _ptr(w_x), _ptr(w_x_b), _ptr(x_dbl), _ptr(x_dbl_b)
    Q.m, Q.tile = m, int(tile)
    Q.wx_row_stride = w_x.stride(0)
    Q.xdbl_batch_stride, Q.xdbl_row_stride = x_dbl.stride(0), x_dbl.stride(1)
    _call("vms_conv_xproj_dual", Q, x)


def conv_bwd(x, weight, bias, dout, dx, dweight, dbias, silu, reverse=False, dx_accumulate=False, reverse_from=0):
    Q = ConvBwdParams()
    fill_conv_fwd(Q.f, x, weight, bias, None, silu, reverse, reverse_from)
    Q.dout, Q.dx, Q.dweight, Q.dbias = _ptr(dout), _ptr(dx), _ptr(dweight), _ptr(dbias)
    Q.dout_batch_stride, Q.dout_c_stride, Q.dout_l_stride = dout.stride()
    Q.dx_batch_stride, Q.dx_c_stride, Q.dx_l_stride = dx.stride()
    Q.dweight_c_stride, Q.dweight_width_stride = dweight.stride()
    Q.dx_accumulate = int(bool(dx_accumulate))
    _call("vms_causal_conv1d_bwd", Q, x)


def conv_update(x, conv_state, weight, bias, out, silu):
    P = ConvFwdParams()
    P.batch, P.dim = x.shape
    P.seqlen, P.width = 1, weight.shape[-1]
    P.dtype, P.wdtype = dtype_code(x), dtype_code(weight)
    P.silu_activation = int(bool(silu))
    P.x, P.weight, P.bias, P.out = _ptr(x), _ptr(weight), _ptr(bias), _ptr(out)
    P.x_batch_stride, P.x_c_stride, P.x_l_stride = x.stride(0), x.stride(1), 1
    P.weight_c_stride, P.weight_width_stride = weight.stride()
    P.out_batch_stride, P.out_c_stride, P.out_l_stride = out.stride(0), out.stride(1), 1
    P.conv_state = _ptr(conv_state)
    P.conv_state_batch_stride, P.conv_state_c_stride, P.conv_state_l_stride = conv_state.stride()
    _call("vms_causal_conv1d_update", P, x)


# ---- fused add + LayerNorm / RMSNorm --------------------------------------------------------------------
def norm_fwd(x, residual, weight, bias, y, residual_out, mean, rstd, eps, is_rms):
    """x, y, residual, residual_out: 2-D, unit column stride; weight / bias: fp32 (cols)."""
    P = NormParams()
    P.rows, P.cols = x.shape
    P.x_dtype = dtype_code(x)
    ref = residual if residual is not None else residual_out
    P.res_dtype = dtype_code(ref) if ref is not None else P.x_dtype
    P.is_rms, P.eps = int(bool(is_rms)), float(eps)
    P.x, P.residual, P.weight, P.bias = _ptr(x), _ptr(residual), _ptr(weight), _ptr(bias)
    P.y, P.residual_out, P.mean, P.rstd = _ptr(y), _ptr(residual_out), _ptr(mean), _ptr(rstd)
    P.x_row_stride, P.y_row_stride = x.stride(0), y.stride(0)
    if residual is not None:
        P.residual_row_stride = residual.stride(0)
    if residual_out is not None:
        P.residual_out_row_stride = residual_out.stride(0)
    _call("vms_layer_norm_fwd", P, x)


def norm_bwd_partials(rows, cols):
    P = NormParams()
    P.rows, P.cols = rows, cols
    return lib().vms_layer_norm_bwd_partials(ctypes.byref(P))


def sum_slices(t, out_dtype):
    """t (n_slices, ...) contiguous, 16-bit -> t.sum(0, dtype=out_dtype) with fp32 accumulation as one streaming launch (vms_hip.h
    vms_sum_slices); anything else (CPU, fp32 slices, odd sizes) goes to torch."""
    n = t[0].numel() if t.dim() > 0 and t.shape[0] > 0 else 0
    if not (t.is_cuda and t.dtype in (torch.bfloat16, torch.float16) and t.is_contiguous() and n > 0 and n % 8 == 0
            and out_dtype in (torch.float32, torch.bfloat16, torch.float16) and t.data_ptr() % 16 == 0):
        return t.sum(0, dtype=out_dtype)
    out = torch.empty(t.shape[1:], dtype=out_dtype, device=t.device)
    _call_plain("vms_sum_slices", t, ctypes.c_void_p(t.data_ptr()), dtype_code(t), int(t.shape[0]), ctypes.c_int64(n), ctypes.c_int64(n),
                ctypes.c_void_p(out.data_ptr()), _DTYPE[out_dtype])
    return out


def norm_bwd_finish(dw_partial, db_partial, dw, db):
    """dw = dw_partial.sum(0) (and db), rounded to dw's dtype: one launch for both arrays (vms_hip.h vms_layer_norm_bwd_finish)"""
    if not dw_partial.is_cuda:
        raise RuntimeError("vms_layer_norm_bwd_finish: tensors must be on a GPU (no CPU path in this library)")
    n_part, cols = dw_partial.shape
    if (dw_partial.dtype != torch.float32 or not dw_partial.is_contiguous() or (db_partial is not None and (
            db_partial.dtype != torch.float32 or not db_partial.is_contiguous() or db_partial.shape != dw_partial.shape))
            or tuple(dw.shape) != (cols,) or not dw.is_contiguous() or (db is not None and (tuple(db.shape) != (cols,) or db.dtype != dw.dtype))):
        raise RuntimeError("norm_bwd_finish: contiguous fp32 (n_partials, cols) partials and (cols,) outputs of one dtype expected")
    _call_plain("vms_layer_norm_bwd_finish", dw_partial, ctypes.c_void_p(_ptr(dw_partial)), ctypes.c_void_p(_ptr(db_partial)), int(n_part), int(cols),
                ctypes.c_void_p(_ptr(dw)), ctypes.c_void_p(_ptr(db)), dtype_code(dw))


def norm_bwd(s, dy, weight, mean, rstd, dres_out, dx, dres_in, dw_partial, db_partial, is_rms):
    Q = NormBwdParams()
    Q.f.rows, Q.f.cols = s.shape
    Q.f.x_dtype, Q.f.res_dtype = dtype_code(dy), dtype_code(s)
    Q.f.is_rms = int(bool(is_rms))
    Q.f.weight, Q.f.mean, Q.f.rstd = _ptr(weight), _ptr(mean), _ptr(rstd)
    Q.s, Q.dy, Q.dres_out, Q.dx, Q.dres_in = _ptr(s), _ptr(dy), _ptr(dres_out), _ptr(dx), _ptr(dres_in)
    Q.dw_partial, Q.db_partial, Q.n_partials = _ptr(dw_partial), _ptr(db_partial), dw_partial.shape[0]
    Q.s_row_stride, Q.dy_row_stride, Q.dx_row_stride = s.stride(0), dy.stride(0), dx.stride(0)
    if dres_out is not None:
        Q.dres_out_row_stride = dres_out.stride(0)
    if dres_in is not None:
        Q.dres_in_row_stride = dres_in.stride(0)
    _call("vms_layer_norm_bwd", Q, s)


# ---- small projections of the inner node (csrc/inner_proj.hip) -----------------------------------------------
def proj_apply(w, inp, out, accumulate=False, tiles_per_wg=0):
    """out[b, d, l] (+)= sum_r w[d, r] inp[b, r, l];  w (rows, k) any strides, inp (batch, k, seqlen), out (batch, rows, seqlen),
    unit seqlen strides, 16-bit dtype (vms_hip.h vms_proj_apply)."""
    P = ProjApplyParams()
    P.batch, P.k, P.seqlen = inp.shape
    P.rows = w.shape[0]
    P.dtype, P.accumulate, P.tiles_per_wg = dtype_code(inp), int(bool(accumulate)), int(tiles_per_wg)
    if w.dtype != inp.dtype or out.dtype != inp.dtype or inp.stride(2) != 1 or out.stride(2) != 1:
        raise RuntimeError("proj_apply: w, inp and out must share one 16-bit dtype; unit seqlen strides")
    if tuple(out.shape) != (inp.shape[0], w.shape[0], inp.shape[2]) or w.shape[1] != inp.shape[1]:
        raise RuntimeError("proj_apply: out must be (batch, rows, seqlen) and w (rows, k)")
    P.w, P.inp, P.out = _ptr(w), _ptr(inp), _ptr(out)
    P.w_row_stride, P.w_k_stride = w.stride(0), w.stride(1)
    P.in_batch_stride, P.in_k_stride = inp.stride(0), inp.stride(1)
    P.out_batch_stride, P.out_row_stride = out.stride(0), out.stride(1)
    _call("vms_proj_apply", P, inp)


def proj_wgrad(p, q, dw, tiles_per_wg=0, transposed=False):
    """dw[m, n] += sum_{b, l} p[b, m, l] q[b, n, l];  dw fp32 (m, n), added to with atomics (vms_hip.h vms_proj_wgrad).
    transposed: dw is the (n, m) matrix instead (dw[n, m] += ...)."""
    P = ProjWgradParams()
    P.batch, P.m, P.seqlen = p.shape
    P.n = q.shape[1]
    P.dtype, P.tiles_per_wg, P.dw_transposed = dtype_code(p), int(tiles_per_wg), int(bool(transposed))
    if q.dtype != p.dtype or dw.dtype != torch.float32 or p.stride(2) != 1 or q.stride(2) != 1 or dw.stride(1) != 1:
        raise RuntimeError("proj_wgrad: p and q must share one 16-bit dtype with unit seqlen strides; dw fp32, unit column stride")
    want = (q.shape[1], p.shape[1]) if transposed else (p.shape[1], q.shape[1])
    if tuple(dw.shape) != want or q.shape[0] != p.shape[0] or q.shape[2] != p.shape[2]:
        raise RuntimeError("proj_wgrad: p (batch, m, seqlen), q (batch, n, seqlen), dw (m, n) -- or (n, m) with transposed -- expected")
    P.p, P.q, P.dw = _ptr(p), _ptr(q), _ptr(dw)
    P.p_batch_stride, P.p_row_stride = p.stride(0), p.stride(1)
    P.q_batch_stride, P.q_row_stride = q.stride(0), q.stride(1)
    P.dw_row_stride = dw.stride(0)
    _call("vms_proj_wgrad", P, p)


def proj_kred_eligible(w, inp, out):
    """Does vms_proj_kred take out[b, m, l] = sum_k w[m, k] inp[b, k, l]?  (16-bit tensors of one dtype, unit seqlen strides, seqlen
    and the strides of inp multiples of 8, 16-byte aligned bases, m <= 96, w contiguous along k or along m in whole 16-byte pieces)"""
    if not (inp.is_cuda and inp.dtype in (torch.bfloat16, torch.float16) and w.dtype == inp.dtype and out.dtype == inp.dtype
            and inp.dim() == 3 and out.dim() == 3 and w.dim() == 2 and inp.stride(2) == 1 and out.stride(2) == 1):
        return False
    m, k = w.shape
    if not (1 <= m <= 96 and tuple(inp.shape[1:2]) == (k,) and tuple(out.shape) == (inp.shape[0], m, inp.shape[2])):
        return False
    if inp.shape[2] % 8 or inp.stride(0) % 8 or inp.stride(1) % 8 or inp.data_ptr() % 16 or w.data_ptr() % 16:
        return False
    if ((k - 1) * inp.stride(1) + inp.shape[2]) * 2 >= 2 ** 31:   # a batch entry is addressed through one buffer resource
        return False
    return (w.stride(1) == 1 and w.stride(0) % 8 == 0 and k % 8 == 0) or (w.stride(0) == 1 and w.stride(1) % 8 == 0 and m % 8 == 0)


def proj_kred(w, inp, out, w2=None, inp2=None, out2=None, tile=0, cast_src=None, cast_src2=None):
    """out[b, m, l] = sum_k w[m, k] inp[b, k, l]  (vms_hip.h vms_proj_kred: x_dbl = x_proj.weight @ conv1d_out, dx_dbl[:R] =
    dt_proj.weight^T @ ddelta); w2 / inp2 / out2: a second problem of the same shape and strides in the same launch.
    cast_src: fp32 (groups, batch, rows, seqlen) -> rounded into the groups * rows rows that follow `out`'s m rows in its parent
    tensor (the backward's dB / dC next to its d_dt: `out` must be the first m rows of a (batch, m + groups * rows, seqlen) tensor)."""
    if not proj_kred_eligible(w, inp, out):
        raise RuntimeError("proj_kred: 16-bit w (m <= 96, k), inp (batch, k, seqlen), out (batch, m, seqlen) of one dtype expected; unit "
                           "seqlen strides, seqlen / strides multiples of 8, 16-byte aligned, w contiguous along k or m")
    P = ProjKredParams()
    P.batch, P.k, P.seqlen = inp.shape
    P.m = w.shape[0]
    P.dtype, P.tile = dtype_code(inp), int(tile)
    P.w, P.inp, P.out = _ptr(w), _ptr(inp), _ptr(out)
    P.w_row_stride, P.w_k_stride = w.stride(0), w.stride(1)
    P.in_batch_stride, P.in_k_stride = inp.stride(0), inp.stride(1)
    P.out_batch_stride, P.out_row_stride = out.stride(0), out.stride(1)
    if w2 is not None:
        if not (proj_kred_eligible(w2, inp2, out2) and w2.shape == w.shape and inp2.shape == inp.shape and w2.stride() == w.stride()
                and inp2.stride() == inp.stride() and out2.stride() == out.stride() and inp2.dtype == inp.dtype):
            raise RuntimeError("proj_kred: the second problem must have the first one's shapes, strides and dtype")
        P.w2, P.inp2, P.out2 = _ptr(w2), _ptr(inp2), _ptr(out2)
    if cast_src is not None:
        if not (cast_src.dtype == torch.float32 and cast_src.dim() == 4 and cast_src.stride(3) == 1 and cast_src.shape[1] == inp.shape[0]
                and cast_src.shape[3] == inp.shape[2]):
            raise RuntimeError("proj_kred: cast_src must be fp32 (groups, batch, rows, seqlen) with a unit seqlen stride")
        P.cast_src, P.cast_groups, P.cast_rows = _ptr(cast_src), cast_src.shape[0], cast_src.shape[2]
        P.cast_group_stride, P.cast_batch_stride, P.cast_row_stride = cast_src.stride(0), cast_src.stride(1), cast_src.stride(2)
        if cast_src2 is not None:
            if cast_src2.shape != cast_src.shape or cast_src2.stride() != cast_src.stride() or cast_src2.dtype != torch.float32:
                raise RuntimeError("proj_kred: cast_src2 must have cast_src's shape and strides")
            P.cast_src2 = _ptr(cast_src2)
    _call("vms_proj_kred", P, inp)


def proj_conv_bwd(x, du, dx_dbl, w_x, conv_w, conv_b, dx, dconv_w, dconv_b, dw_x, reverse=False, reverse_from=0,
                  dx_accumulate=False, tiles_per_wg=0):
    """dw_x += dx_dbl conv1d_out^T;  dx, dconv_w, dconv_b = conv1d backward of (du + w_x^T dx_dbl), SiLU on
    (vms_hip.h vms_proj_conv_bwd).  x, du, dx: (batch, dim, seqlen); dx_dbl: (batch, k, seqlen); w_x: (k, dim);
    conv_w: (dim, width); dconv_w / dconv_b / dw_x: fp32, zero-filled by the caller."""
    P = ProjConvBwdParams()
    P.batch, P.dim, P.seqlen = x.shape
    P.k, P.width = dx_dbl.shape[1], conv_w.shape[1]
    P.dtype, P.wdtype = dtype_code(x), dtype_code(conv_w)
    P.reverse, P.reverse_from, P.dx_accumulate, P.tiles_per_wg = int(bool(reverse)), int(reverse_from), int(bool(dx_accumulate)), int(tiles_per_wg)
    for t, name in ((x, "x"), (du, "du"), (dx_dbl, "dx_dbl"), (dx, "dx")):
        if t.dtype != x.dtype or t.stride(2) != 1:
            raise RuntimeError(f"proj_conv_bwd: {name} must have x's dtype and a unit seqlen stride")
    if w_x.dtype != x.dtype or tuple(w_x.shape) != (P.k, P.dim) or tuple(du.shape) != tuple(x.shape) or tuple(dx.shape) != tuple(x.shape):
        raise RuntimeError("proj_conv_bwd: w_x must be (k, dim) in x's dtype; du and dx shaped like x")
    if dw_x.dtype != torch.float32 or dconv_w.dtype != torch.float32 or dw_x.stride(1) != 1 or tuple(dw_x.shape) != (P.k, P.dim):
        raise RuntimeError("proj_conv_bwd: dw_x (k, dim) and dconv_w (dim, width) must be float32 accumulators")
    if conv_b is not None and conv_b.dtype != conv_w.dtype:
        raise RuntimeError("proj_conv_bwd: conv bias must have the conv weight's dtype")
    P.x, P.du, P.dx_dbl, P.w_x, P.conv_weight, P.conv_bias = _ptr(x), _ptr(du), _ptr(dx_dbl), _ptr(w_x), _ptr(conv_w), _ptr(conv_b)
    P.dx, P.dconv_weight, P.dconv_bias, P.dw_x = _ptr(dx), _ptr(dconv_w), _ptr(dconv_b), _ptr(dw_x)
    P.x_batch_stride, P.x_c_stride = x.stride(0), x.stride(1)
    P.du_batch_stride, P.du_c_stride = du.stride(0), du.stride(1)
    P.dxdbl_batch_stride, P.dxdbl_k_stride = dx_dbl.stride(0), dx_dbl.stride(1)
    P.wx_k_stride, P.wx_c_stride = w_x.stride(0), w_x.stride(1)
    P.conv_weight_c_stride, P.conv_weight_width_stride = conv_w.stride(0), conv_w.stride(1)
    P.dx_batch_stride, P.dx_c_stride = dx.stride(0), dx.stride(1)
    P.dconv_weight_c_stride, P.dconv_weight_width_stride = dconv_w.stride(0), dconv_w.stride(1)
    P.dwx_k_stride = dw_x.stride(0)
    _call("vms_proj_conv_bwd", P, x)


# ---- single-token SSM step ---------------------------------------------------------------------------------
def state_update(state, x, dt, A, B, C, D, z, dt_bias, out, dt_softplus):
    P = StateUpdateParams()
    P.batch, P.dim, P.dstate = state.shape
    P.state_dtype, P.x_dtype, P.bc_dtype, P.w_dtype = dtype_code(state), dtype_code(x), dtype_code(B), dtype_code(A)
    P.dt_dtype = dtype_code(dt)
    P.z_dtype = dtype_code(z) if z is not None else P.x_dtype
    P.dt_softplus = int(bool(dt_softplus))
    P.state, P.x, P.dt, P.A, P.B, P.C = _ptr(state), _ptr(x), _ptr(dt), _ptr(A), _ptr(B), _ptr(C)
    P.D, P.z, P.dt_bias, P.out = _ptr(D), _ptr(z), _ptr(dt_bias), _ptr(out)
    P.state_batch_stride, P.state_d_stride, P.state_n_stride = state.stride()
    P.x_batch_stride, P.x_d_stride = x.stride()
    P.dt_batch_stride, P.dt_d_stride = dt.stride()
    if z is not None:
        P.z_batch_stride, P.z_d_stride = z.stride()
    P.out_batch_stride, P.out_d_stride = out.stride()
    P.A_d_stride, P.A_n_stride = A.stride()
    P.B_batch_stride, P.B_n_stride = B.stride()
    P.C_batch_stride, P.C_n_stride = C.stride()
    _call("vms_selective_state_update", P, x)


class PrepPlan:
    """vms_param_prep for a FIXED list of jobs whose destinations move every step (a block's per-step weight preparation): the
    parameter block is filled once -- sources, shapes, strides, dtypes -- and a launch only rewrites the destination pointers
    (param_prep builds it from tensors every time: ~45 us of host time per block step).
    jobs: (src tensor, dst group, dst byte offset, dst shape, dst strides in elements, dst dtype, op); the destinations are
    regions of `n_groups` buffers whose base pointers the caller passes to run()."""

    def __init__(self, jobs):
        assert 0 < len(jobs) <= PREP_MAX_JOBS
        self.P = PrepParams()
        self.P.n_jobs = len(jobs)
        # what a later call must still find: the sources' addresses AND their geometry (a parameter re-viewed or resized in place at
        # the same address -- p.data = other_view, resize_ -- must not run on the stale rows / strides: ADVICE r4)
        self.src_key = tuple(self._key(src) for src, *_ in jobs)
        self.dst = []
        codes = {torch.float32: VMS_F32, torch.float16: VMS_F16, torch.bfloat16: VMS_BF16}
        for j, (src, group, off, dshape, dstrides, ddtype, op) in zip(self.P.job, jobs):
            s2 = src if src.dim() == 2 else src.reshape(1, -1)
            dshape, dstrides = (tuple(dshape), tuple(dstrides)) if len(dshape) == 2 else ((1, dshape[0]), (dshape[0], dstrides[0]))
            if s2.stride(1) != 1 or dstrides[1] < 1 or dshape != (tuple(s2.shape) if op != PREP_CAST_T else tuple(s2.shape[::-1])):
                raise RuntimeError("param_prep: jobs are (rows, cols) matrices, src with a unit column stride, and matching shapes")
            j.src, j.rows, j.cols = _ptr(s2), s2.shape[0], s2.shape[1]
            j.dst_col_stride = dstrides[1] if dshape[1] > 1 else 1
            j.src_row_stride = s2.stride(0) if s2.shape[0] > 1 else s2.shape[1]
            j.dst_row_stride = dstrides[0] if dshape[0] > 1 else dshape[1]
            j.src_dtype, j.dst_dtype, j.op = dtype_code(s2), codes[ddtype], op
            self.dst.append((group, off))

    @staticmethod
    def _key(t):
        return (t.data_ptr(), tuple(t.shape), tuple(t.stride()), t.dtype)

    def matches(self, srcs):
        return self.src_key == tuple(self._key(t) for t in srcs)

    def run(self, bases, ref_tensor):
        # every launch gets its OWN copy of the block with this step's destinations: two threads running the same module (two
        # streams of inference) used to rewrite the one cached block under each other.  ~0.5 KB memmove.
        P = PrepParams()
        ctypes.memmove(ctypes.byref(P), ctypes.byref(self.P), ctypes.sizeof(PrepParams))
        for j, (group, off) in zip(P.job, self.dst):
            j.dst = bases[group] + off
        _call("vms_param_prep", P, ref_tensor)


def param_prep(jobs):
    """jobs: up to PREP_MAX_JOBS triples (src, dst, op) of 2-D tensors (1-D ones count as one row), src with a unit column stride, dst
    with any column stride (2 = one half of an interleaved matrix):
    PREP_CAST dst = src in dst's dtype, PREP_CAST_T dst = src^T, PREP_NEG_EXP dst = -exp(src); one launch (vms_hip.h)."""
    P = PrepParams()
    assert 0 < len(jobs) <= PREP_MAX_JOBS
    P.n_jobs = len(jobs)
    for j, (src, dst, op) in zip(P.job, jobs):
        s2 = src if src.dim() == 2 else src.reshape(1, -1)
        d2 = dst if dst.dim() == 2 else dst.reshape(1, -1)
        if s2.stride(1) != 1 or d2.stride(1) < 1 or tuple(d2.shape) != (tuple(s2.shape) if op != PREP_CAST_T else tuple(s2.shape[::-1])):
            raise RuntimeError("param_prep: jobs are (rows, cols) matrices, src with a unit column stride, and matching shapes")
        j.src, j.dst, j.rows, j.cols = _ptr(s2), _ptr(d2), s2.shape[0], s2.shape[1]
        j.dst_col_stride = d2.stride(1) if d2.shape[1] > 1 else 1
        j.src_row_stride, j.dst_row_stride = s2.stride(0) if s2.shape[0] > 1 else s2.shape[1], d2.stride(0) if d2.shape[0] > 1 else d2.shape[1]
        j.src_dtype, j.dst_dtype, j.op = dtype_code(s2), dtype_code(d2), op
    _call("vms_param_prep", P, jobs[0][0])
