// vms_torch.cpp -- compiled binding of the C ABI (include/vms_hip.h) for PyTorch: the module the Python mirrors
// selective_scan_cuda.py / causal_conv1d_cuda.py forward to when it has been built.
//
// It plays the part of the reference's two pybind11 host files
//   mamba/csrc/selective_scan/selective_scan.cpp:226-497        (fwd, bwd)
//   causal-conv1d/csrc/causal_conv1d.cpp:130-333                 (causal_conv1d_fwd, _bwd, _update)
// checks (TORCH_CHECK -> RuntimeError, same expressions), output allocation rules (out = empty_like(delta), fp32 dB /
// dC cast on return, caller-owned dz / dx), parameter-block fill, device guard + current stream -- all in C++, so that
// a launch costs microseconds of host time instead of the ~40-60 us of the ctypes path (what small problems such as
// the DBM block at (2, 2304, 512) are bound by).  Nothing here computes: the kernels live in libvms_hip.so.
#include <torch/extension.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <limits>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "../../../include/vms_hip.h"

namespace {

using torch::Tensor;
using OptT = c10::optional<Tensor>;

int dtype_code(const Tensor& t) {
    switch (t.scalar_type()) {
        case at::kFloat: return VMS_F32;
        case at::kHalf: return VMS_F16;
        case at::kBFloat16: return VMS_BF16;
        default: TORCH_CHECK(false, "unsupported dtype ", t.scalar_type(), ": expected float32, float16 or bfloat16");
    }
    return 0;
}
bool is_itype(const Tensor& t) {
    return t.scalar_type() == at::kFloat || t.scalar_type() == at::kHalf || t.scalar_type() == at::kBFloat16;
}
const void* cptr(const OptT& t) { return t.has_value() ? t->data_ptr() : nullptr; }
void* mptr(const OptT& t) { return t.has_value() ? t->data_ptr() : nullptr; }

// ---- optional per-launch device timing (bench.py): two events per C-ABI call on the launch stream ---------------
// Calls arrive from the Python thread AND from autograd's backward threads: the flag is atomic, the record list and the
// pool of pre-created events are guarded by one mutex (held only around list / pool operations, never around a launch).
struct Timed { const char* name; hipEvent_t e0, e1; const char* kernel; };   // kernel: vms_last_kernel() behind the launch
std::atomic<bool> g_timing{false};
std::mutex g_timing_mu;
std::string g_timing_only;   // non-empty: only this entry point is timed (written under g_timing_mu before g_timing is raised)
std::vector<Timed> g_timed;
std::vector<hipEvent_t> g_event_pool;

hipEvent_t take_event_locked() {
    if (!g_event_pool.empty()) {
        hipEvent_t e = g_event_pool.back();
        g_event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

template <typename P>
void call(const char* name, int (*fn)(const P*, void*), const P& p, const Tensor& ref) {
    TORCH_CHECK(ref.is_cuda(), name, ": tensors must be on a GPU (no CPU path in this library)");
    c10::DeviceGuard guard(ref.device());
    hipStream_t s = c10::hip::getCurrentHIPStream(ref.device().index()).stream();
    int rc;
    if (g_timing.load(std::memory_order_acquire) && (g_timing_only.empty() || g_timing_only == name)) {
        Timed t{name, nullptr, nullptr, ""};
        {
            std::lock_guard<std::mutex> lk(g_timing_mu);
            t.e0 = take_event_locked();
            t.e1 = take_event_locked();
        }
        (void)hipEventRecord(t.e0, s);
        rc = fn(&p, s);
        (void)hipEventRecord(t.e1, s);
        t.kernel = vms_last_kernel();   // thread-local in the library, a string literal: what this very launch chose
        std::lock_guard<std::mutex> lk(g_timing_mu);
        g_timed.push_back(t);
    } else {
        rc = fn(&p, s);
    }
    TORCH_CHECK(rc == 0, name, " failed (status ", rc, "): ", vms_last_error());
}

// ---- selective scan -------------------------------------------------------------------------------------------------
struct ScanDims { int64_t batch, dim, seqlen, dstate; bool var_B, var_C; };

ScanDims scan_checks(const Tensor& u, const Tensor& delta, const Tensor& A, const Tensor& B, const Tensor& C, const OptT& D_,
                     const OptT& z_, const OptT& delta_bias_) {   // selective_scan.cpp:233-305
    TORCH_CHECK(is_itype(u), "selective_scan: input dtype ", u.scalar_type(), " not supported");
    TORCH_CHECK(!A.is_complex(), "selective_scan: complex A runs through the ctypes binding (selective_scan_cuda.py routes it there)");
    TORCH_CHECK(A.scalar_type() == at::kFloat, "selective_scan: A must be float32");
    const bool var_B = B.dim() >= 3, var_C = C.dim() >= 3;
    TORCH_CHECK(delta.scalar_type() == u.scalar_type(), "delta.scalar_type() == input_type");
    TORCH_CHECK(B.scalar_type() == (var_B ? u.scalar_type() : A.scalar_type()), "B.scalar_type() == (!is_variable_B ? weight_type : input_type)");
    TORCH_CHECK(C.scalar_type() == (var_C ? u.scalar_type() : A.scalar_type()), "C.scalar_type() == (!is_variable_C ? weight_type : input_type)");
    TORCH_CHECK(u.is_cuda(), "u.is_cuda()");
    TORCH_CHECK(delta.is_cuda(), "delta.is_cuda()");
    TORCH_CHECK(A.is_cuda(), "A.is_cuda()");
    TORCH_CHECK(B.is_cuda(), "B.is_cuda()");
    TORCH_CHECK(C.is_cuda(), "C.is_cuda()");
    TORCH_CHECK(u.dim() == 3, "u must be (batch, dim, seqlen)");
    TORCH_CHECK(u.stride(-1) == 1, "u.stride(-1) == 1");
    TORCH_CHECK(delta.stride(-1) == 1, "delta.stride(-1) == 1");
    const int64_t batch = u.size(0), dim = u.size(1), seqlen = u.size(2), dstate = A.size(1);
    const int64_t n_groups = var_B ? B.size(1) : 1;
    TORCH_CHECK(dstate <= 256, "selective_scan only supports state dimension <= 256");
    TORCH_CHECK(delta.dim() == 3 && delta.size(0) == batch && delta.size(1) == dim && delta.size(2) == seqlen, "delta must have shape (batch, dim, seqlen)");
    TORCH_CHECK(A.dim() == 2 && A.size(0) == dim, "A must have shape (dim, dstate)");
    if (!var_B) {
        TORCH_CHECK(B.dim() == 2 && B.size(0) == dim && B.size(1) == dstate, "B must have shape (dim, dstate)");
    } else {
        TORCH_CHECK(B.dim() == 4 && B.size(0) == batch && B.size(2) == dstate && B.size(3) == seqlen, "B must have shape (batch, n_groups, dstate, seqlen)");
        TORCH_CHECK(B.stride(-1) == 1, "B.stride(-1) == 1");
    }
    if (!var_C) {
        TORCH_CHECK(C.dim() == 2 && C.size(0) == dim && C.size(1) == dstate, "C must have shape (dim, dstate)");
    } else {
        TORCH_CHECK(C.dim() == 4 && C.size(0) == batch && C.size(2) == dstate && C.size(3) == seqlen && (!var_B || C.size(1) == n_groups),
                    "C must have shape (batch, n_groups, dstate, seqlen)");
        TORCH_CHECK(C.stride(-1) == 1, "C.stride(-1) == 1");
    }
    auto vec_check = [&](const OptT& t, const char* name) {
        if (!t.has_value()) return;
        TORCH_CHECK(t->scalar_type() == at::kFloat, name, " must be float32");
        TORCH_CHECK(t->is_cuda(), name, ".is_cuda()");
        TORCH_CHECK(t->stride(-1) == 1, name, ".stride(-1) == 1");
        TORCH_CHECK(t->dim() == 1 && t->size(0) == dim, name, " must have shape (dim,)");
    };
    vec_check(D_, "D");
    vec_check(delta_bias_, "delta_bias");
    if (z_.has_value()) {
        TORCH_CHECK(z_->scalar_type() == u.scalar_type(), "z.scalar_type() == input_type");
        TORCH_CHECK(z_->is_cuda(), "z.is_cuda()");
        TORCH_CHECK(z_->stride(-1) == 1, "z.stride(-1) == 1");
        TORCH_CHECK(z_->dim() == 3 && z_->size(0) == batch && z_->size(1) == dim && z_->size(2) == seqlen, "z must have shape (batch, dim, seqlen)");
    }
    return {batch, dim, seqlen, dstate, var_B, var_C};
}

void fill_scan(vms_scan_fwd_params& P, const ScanDims& s, const Tensor& u, const Tensor& delta, const Tensor& A, const Tensor& B,
               const Tensor& C, const OptT& D_, const OptT& z_, const OptT& delta_bias_, const OptT& out, const OptT& out_z,
               const OptT& x, bool delta_softplus, bool reverse, int64_t impl, int64_t segments, int64_t bc_pad, int64_t reverse_from = 0) {
    P = vms_scan_fwd_params{};
    TORCH_CHECK(reverse_from >= 0 && reverse_from <= s.batch && (reverse_from == 0 || !reverse), "reverse_from must be in [0, batch] with reverse = False");
    P.reverse_from = (int)reverse_from;
    P.batch = (int)s.batch; P.dim = (int)s.dim; P.seqlen = (int)s.seqlen; P.dstate = (int)s.dstate;
    P.n_groups = s.var_B ? (int)B.size(1) : (s.var_C ? (int)C.size(1) : 1);
    P.n_chunks = (int)((s.seqlen + 2047) / 2048);
    P.dtype = dtype_code(u);
    P.is_variable_B = s.var_B; P.is_variable_C = s.var_C; P.delta_softplus = delta_softplus;
    P.reverse = reverse; P.impl = (int)impl; P.segments = (int)segments; P.bc_pad = (int)bc_pad;
    P.u = u.data_ptr(); P.delta = delta.data_ptr(); P.A = A.data_ptr(); P.B = B.data_ptr(); P.C = C.data_ptr();
    P.D = cptr(D_); P.z = cptr(z_); P.delta_bias = cptr(delta_bias_);
    P.out = mptr(out); P.out_z = mptr(out_z); P.x = mptr(x);
    if (x.has_value()) {
        P.x_chunk_stride = x->stride(2);
        P.x_has_sub = x->stride(3) != 1 ? 0 : x->stride(2) >= 258 * s.dstate ? 3 : x->stride(2) >= 18 * s.dstate ? 1 : 0;
    }
    P.u_batch_stride = u.stride(0); P.u_d_stride = u.stride(1);
    P.delta_batch_stride = delta.stride(0); P.delta_d_stride = delta.stride(1);
    if (z_.has_value()) { P.z_batch_stride = z_->stride(0); P.z_d_stride = z_->stride(1); }
    if (out.has_value()) { P.out_batch_stride = out->stride(0); P.out_d_stride = out->stride(1); }
    if (out_z.has_value()) { P.out_z_batch_stride = out_z->stride(0); P.out_z_d_stride = out_z->stride(1); }
    P.A_d_stride = A.stride(0); P.A_dstate_stride = A.stride(1);
    if (s.var_B) { P.B_batch_stride = B.stride(0); P.B_group_stride = B.stride(1); P.B_dstate_stride = B.stride(2); }
    else { P.B_d_stride = B.stride(0); P.B_dstate_stride = B.stride(1); }
    if (s.var_C) { P.C_batch_stride = C.stride(0); P.C_group_stride = C.stride(1); P.C_dstate_stride = C.stride(2); }
    else { P.C_d_stride = C.stride(0); P.C_dstate_stride = C.stride(1); }
}

// -> [out, x, (out_z)]   (selective_scan.cpp:226-336).  B / C must already carry the padding bc_pad promises.
std::vector<Tensor> scan_fwd(const Tensor& u, const Tensor& delta, const Tensor& A, const Tensor& B, const Tensor& C, const OptT& D_,
                             const OptT& z_, const OptT& delta_bias_, bool delta_softplus, bool reverse, const OptT& out_z_into,
                             int64_t bc_pad, int64_t impl, int64_t segments, int64_t reverse_from = 0, int64_t x_mode = -1) {
    const ScanDims s = scan_checks(u, delta, A, B, C, D_, z_, delta_bias_);
    // before the workspace query and the allocations: the split decision reads the CU count of the CURRENT device
    c10::DeviceGuard guard(u.device());
    Tensor out = at::empty_like(delta);   // inherits delta's (d-slowest) layout, selective_scan.cpp:310-311
    OptT out_z;
    if (z_.has_value()) out_z = at::empty_like(*z_);
    if (out_z_into.has_value()) {
        TORCH_CHECK(z_.has_value(), "out_z_into needs z");
        TORCH_CHECK(out_z_into->scalar_type() == u.scalar_type() && out_z_into->is_cuda() && out_z_into->stride(-1) == 1 &&
                        out_z_into->dim() == 3 && out_z_into->size(0) == s.batch && out_z_into->size(1) == s.dim && out_z_into->size(2) == s.seqlen,
                    "out_z_into must be (batch, dim, seqlen), input dtype, unit last stride");
        out_z = out_z_into;
    }
    const int64_t n_chunks = (s.seqlen + 2047) / 2048;
    // the reference-shaped x is a view of a wider buffer whose tail carries 128-element sub-checkpoints for the
    // backward kernel (include/vms_hip.h)
    // (the library picks the pitch: 8-element checkpoints -- x_has_sub == 3 -- when the backward kernel that reads them takes the
    // problem and x_mode allows, else the 128-element ones; only sizes and flags of the descriptor are read)
    vms_scan_fwd_params P;
    fill_scan(P, s, u, delta, A, B, C, D_, z_, delta_bias_, out, out_z, OptT(), delta_softplus, reverse, impl, segments, bc_pad, reverse_from);
    const int64_t pitch = vms_scan_x_pitch(&P, (int32_t)x_mode);
    Tensor x = at::empty({s.batch, s.dim, n_chunks, pitch}, u.options().dtype(at::kFloat)).narrow(3, 0, s.dstate * 2);
    fill_scan(P, s, u, delta, A, B, C, D_, z_, delta_bias_, out, out_z, x, delta_softplus, reverse, impl, segments, bc_pad, reverse_from);
    P.out_z_accumulate = out_z_into.has_value();
    Tensor ws;
    const int64_t nws = vms_scan_fwd_workspace_bytes(&P);   // state carries of a sequence-split forward
    if (nws > 0) {
        ws = at::empty({nws}, u.options().dtype(at::kByte));
        P.workspace = ws.data_ptr(); P.workspace_bytes = nws;
    }
    call("vms_selective_scan_fwd", vms_selective_scan_fwd, P, u);
    std::vector<Tensor> res{out, x};
    if (z_.has_value()) res.push_back(*out_z);
    return res;
}

// everything selective_scan.cpp:338-492 does before its launch: checks, allocations, the parameter block
struct ScanBwdJob {
    vms_scan_bwd_params Q;
    Tensor du, ddelta, dA, dB, dC, ws, Bshape, Cshape;
    OptT dD, dbias, dz, out_z;
    bool has_z = false, recompute_out_z = false, keep_fp32 = false;
    std::vector<OptT> results() {
        if (!keep_fp32) { dB = dB.to(Bshape.scalar_type()); dC = dC.to(Cshape.scalar_type()); }
        std::vector<OptT> res{du, ddelta, dA, dB, dC, dD, dbias};
        if (has_z) res.push_back(dz);
        if (recompute_out_z) res.push_back(out_z);
        return res;
    }
};

// no_dz: z is given but this call does not produce dz (the second direction of vms_selective_scan_bwd_dual)
ScanBwdJob scan_bwd_prep(const Tensor& u, const Tensor& delta, const Tensor& A, const Tensor& B, const Tensor& C, const OptT& D_,
                         const OptT& z_, const OptT& delta_bias_, const Tensor& dout, const OptT& x_, const OptT& out_, const OptT& dz_,
                         bool delta_softplus, bool recompute_out_z, bool reverse, const OptT& zeroed, bool keep_fp32,
                         bool accumulate_dz, int64_t bc_pad, int64_t impl, int64_t segments, const Tensor& Bshape, const Tensor& Cshape,
                         int64_t reverse_from, bool no_dz = false) {
    const ScanDims s = scan_checks(u, delta, A, B, C, D_, z_, delta_bias_);
    ScanBwdJob J;
    J.Bshape = Bshape; J.Cshape = Cshape; J.keep_fp32 = keep_fp32; J.has_z = z_.has_value(); J.recompute_out_z = recompute_out_z && z_.has_value();
    auto bdl = [&](const Tensor& t) { return t.dim() == 3 && t.size(0) == s.batch && t.size(1) == s.dim && t.size(2) == s.seqlen; };
    TORCH_CHECK(dout.scalar_type() == u.scalar_type(), "dout.scalar_type() == input_type");
    TORCH_CHECK(dout.is_cuda(), "dout.is_cuda()");
    TORCH_CHECK(dout.stride(-1) == 1, "dout.stride(-1) == 1");
    TORCH_CHECK(bdl(dout), "dout must have shape (batch, dim, seqlen)");
    OptT out;
    if (z_.has_value()) {
        TORCH_CHECK(out_.has_value(), "out_.has_value()");
        out = out_;
        TORCH_CHECK(out->scalar_type() == u.scalar_type() && out->is_cuda() && out->stride(-1) == 1 && bdl(*out),
                    "out must be (batch, dim, seqlen), input dtype, unit last stride");
        TORCH_CHECK(!accumulate_dz || dz_.has_value(), "accumulate_dz needs the dz tensor to add to");
        if (dz_.has_value()) {
            J.dz = dz_;
            TORCH_CHECK(J.dz->scalar_type() == u.scalar_type() && J.dz->is_cuda() && J.dz->stride(-1) == 1 && bdl(*J.dz),
                        "dz must be (batch, dim, seqlen), input dtype, unit last stride");
        } else if (!no_dz) {
            J.dz = at::empty_like(*z_);
        }
        if (recompute_out_z) J.out_z = at::empty_like(*out);
    }
    const int64_t n_chunks = (s.seqlen + 2047) / 2048;
    if (n_chunks > 1) TORCH_CHECK(x_.has_value(), "x_.has_value()");
    if (x_.has_value()) {
        TORCH_CHECK(x_->scalar_type() == at::kFloat && x_->is_cuda() && x_->dim() == 4 && x_->size(0) == s.batch && x_->size(1) == s.dim &&
                        x_->size(2) == n_chunks && x_->size(3) == 2 * s.dstate && x_->stride(3) == 1 &&
                        x_->stride(1) == n_chunks * x_->stride(2) && x_->stride(0) == s.dim * x_->stride(1),
                    "x must be the (batch, dim, n_chunks, 2*dstate) checkpoint tensor returned by fwd");
    } else {
        TORCH_CHECK(s.seqlen <= 1024, "x (the forward's checkpoints) is required when seqlen > 1024");
    }
    J.du = at::empty_like(u); J.ddelta = at::empty_like(delta);
    const auto f32 = u.options().dtype(at::kFloat);
    if (zeroed.has_value()) {
        int64_t need = A.numel() + Bshape.numel() + Cshape.numel() + (D_.has_value() ? D_->numel() : 0) + (delta_bias_.has_value() ? delta_bias_->numel() : 0);
        TORCH_CHECK(zeroed->scalar_type() == at::kFloat && zeroed->is_cuda() && zeroed->dim() == 1 && zeroed->is_contiguous() && zeroed->numel() >= need,
                    "zeroed must be a flat float32 tensor of at least bwd_accumulator_elems() elements");
        int64_t o = 0;
        auto carve = [&](const Tensor& like) { Tensor t = zeroed->narrow(0, o, like.numel()).view(like.sizes()); o += like.numel(); return t; };
        J.dA = carve(A); J.dB = carve(Bshape); J.dC = carve(Cshape);
        if (D_.has_value()) J.dD = carve(*D_);
        if (delta_bias_.has_value()) J.dbias = carve(*delta_bias_);
    } else {
        J.dA = at::zeros_like(A);
        J.dB = at::zeros(Bshape.sizes(), f32);
        J.dC = at::zeros(Cshape.sizes(), f32);
        if (D_.has_value()) J.dD = at::zeros_like(*D_);
        if (delta_bias_.has_value()) J.dbias = at::zeros_like(*delta_bias_);
    }
    vms_scan_bwd_params& Q = J.Q;
    Q = vms_scan_bwd_params{};
    fill_scan(Q.f, s, u, delta, A, B, C, D_, z_, delta_bias_, out, J.out_z, x_, delta_softplus, reverse, impl, segments, bc_pad, reverse_from);
    Q.dout = dout.data_ptr(); Q.du = J.du.data_ptr(); Q.ddelta = J.ddelta.data_ptr(); Q.dz = mptr(J.dz);
    Q.dA = J.dA.data_ptr<float>(); Q.dB = J.dB.data_ptr<float>(); Q.dC = J.dC.data_ptr<float>();
    Q.dD = J.dD.has_value() ? J.dD->data_ptr<float>() : nullptr;
    Q.ddelta_bias = J.dbias.has_value() ? J.dbias->data_ptr<float>() : nullptr;
    Q.dout_batch_stride = dout.stride(0); Q.dout_d_stride = dout.stride(1);
    Q.du_batch_stride = J.du.stride(0); Q.du_d_stride = J.du.stride(1);
    Q.ddelta_batch_stride = J.ddelta.stride(0); Q.ddelta_d_stride = J.ddelta.stride(1);
    if (J.dz.has_value()) { Q.dz_batch_stride = J.dz->stride(0); Q.dz_d_stride = J.dz->stride(1); }
    Q.dA_d_stride = J.dA.stride(0); Q.dA_dstate_stride = J.dA.stride(1);
    if (s.var_B) { Q.dB_batch_stride = J.dB.stride(0); Q.dB_group_stride = J.dB.stride(1); Q.dB_dstate_stride = J.dB.stride(2); }
    else { Q.dB_d_stride = J.dB.stride(0); Q.dB_dstate_stride = J.dB.stride(1); }
    if (s.var_C) { Q.dC_batch_stride = J.dC.stride(0); Q.dC_group_stride = J.dC.stride(1); Q.dC_dstate_stride = J.dC.stride(2); }
    else { Q.dC_d_stride = J.dC.stride(0); Q.dC_dstate_stride = J.dC.stride(1); }
    Q.dz_accumulate = accumulate_dz;
    return J;
}

void scan_bwd_workspace(ScanBwdJob& J, const Tensor& u) {   // adjoint carries of a sequence-split backward
    const int64_t nws = vms_scan_bwd_workspace_bytes(&J.Q);
    if (nws > 0) {
        J.ws = at::empty({nws}, u.options().dtype(at::kByte));
        J.Q.f.workspace = J.ws.data_ptr(); J.Q.f.workspace_bytes = nws;
    }
}

// -> [du, ddelta, dA, dB, dC, dD, ddelta_bias, (dz), (out_z)]   (selective_scan.cpp:338-492); B / C as for scan_fwd;
// Bshape / Cshape: the caller's unpadded B / C, whose shapes and dtypes dB / dC take
std::vector<OptT> scan_bwd(const Tensor& u, const Tensor& delta, const Tensor& A, const Tensor& B, const Tensor& C, const OptT& D_,
                           const OptT& z_, const OptT& delta_bias_, const Tensor& dout, const OptT& x_, const OptT& out_, const OptT& dz_,
                           bool delta_softplus, bool recompute_out_z, bool reverse, const OptT& zeroed, bool keep_fp32,
                           bool accumulate_dz, int64_t bc_pad, int64_t impl, int64_t segments, const Tensor& Bshape, const Tensor& Cshape,
                           int64_t reverse_from = 0) {
    c10::DeviceGuard guard(u.device());   // covers the workspace query (CU count of the tensors' device) and the allocations
    ScanBwdJob J = scan_bwd_prep(u, delta, A, B, C, D_, z_, delta_bias_, dout, x_, out_, dz_, delta_softplus, recompute_out_z, reverse, zeroed,
                                 keep_fp32, accumulate_dz, bc_pad, impl, segments, Bshape, Cshape, reverse_from);
    scan_bwd_workspace(J, u);
    call("vms_selective_scan_bwd", vms_selective_scan_bwd, J.Q, u);
    return J.results();
}

// Two launches' worth of work behind one entry point (vms_selective_scan_bwd_dual); timed under the single call's name so that
// per-kernel reports keep one row per entry point
void call_dual(const vms_scan_bwd_params& a, const vms_scan_bwd_params& b, const Tensor& ref) {
    struct Pair { vms_scan_bwd_params a, b; } pr{a, b};
    call<Pair>("vms_selective_scan_bwd_dual",
               [](const Pair* q, void* s) { return vms_selective_scan_bwd_dual(&q->a, &q->b, s); }, pr, ref);
}

// Both directions of a bidirectional block: a (left-to-right) and b (right-to-left) share z, dout and dz.
// -> [results of a (dz = the gradient of z through BOTH directions), results of b without dz]
std::vector<std::vector<OptT>> scan_bwd_dual(
    const Tensor& u_a, const Tensor& delta_a, const Tensor& A_a, const Tensor& B_a, const Tensor& C_a, const OptT& D_a, const OptT& bias_a,
    const OptT& x_a, const Tensor& out_a, const OptT& zeroed_a, const Tensor& Bshape_a, const Tensor& Cshape_a,
    const Tensor& u_b, const Tensor& delta_b, const Tensor& A_b, const Tensor& B_b, const Tensor& C_b, const OptT& D_b, const OptT& bias_b,
    const OptT& x_b, const Tensor& out_b, const OptT& zeroed_b, const Tensor& Bshape_b, const Tensor& Cshape_b,
    const Tensor& z, const Tensor& dout, const OptT& dz_, bool delta_softplus, bool keep_fp32, bool accumulate_dz,
    int64_t bc_pad_a, int64_t bc_pad_b, int64_t impl, int64_t segments) {
    c10::DeviceGuard guard(u_a.device());
    ScanBwdJob Ja = scan_bwd_prep(u_a, delta_a, A_a, B_a, C_a, D_a, z, bias_a, dout, x_a, out_a, dz_, delta_softplus, false, /*reverse=*/false,
                                  zeroed_a, keep_fp32, accumulate_dz, bc_pad_a, impl, segments, Bshape_a, Cshape_a, 0);
    ScanBwdJob Jb = scan_bwd_prep(u_b, delta_b, A_b, B_b, C_b, D_b, z, bias_b, dout, x_b, out_b, OptT(), delta_softplus, false, /*reverse=*/true,
                                  zeroed_b, keep_fp32, false, bc_pad_b, impl, segments, Bshape_b, Cshape_b, 0, /*no_dz=*/true);
    if (!vms_scan_bwd_dual_fused(&Ja.Q, &Jb.Q)) { scan_bwd_workspace(Ja, u_a); scan_bwd_workspace(Jb, u_b); }
    call_dual(Ja.Q, Jb.Q, u_a);
    std::vector<OptT> rb = Jb.results();
    rb.pop_back();   // b has no dz of its own
    return {Ja.results(), rb};
}

// ---- causal conv1d ----------------------------------------------------------------------------------------------------
void conv_common(const Tensor& x, const Tensor& weight, const OptT& bias_) {   // causal_conv1d.cpp:136-170
    TORCH_CHECK(is_itype(x), "causal_conv1d: input dtype ", x.scalar_type(), " not supported");
    TORCH_CHECK(is_itype(weight), "causal_conv1d: weight dtype ", weight.scalar_type(), " not supported");
    TORCH_CHECK(x.is_cuda(), "x.is_cuda()");
    TORCH_CHECK(weight.is_cuda(), "weight.is_cuda()");
    const int64_t width = weight.size(-1);
    TORCH_CHECK(weight.dim() == 2 && weight.size(0) == x.size(1), "weight must have shape (dim, width)");
    TORCH_CHECK(width >= 2 && width <= 4, "causal_conv1d only supports width between 2 and 4");
    if (bias_.has_value()) {
        TORCH_CHECK(bias_->scalar_type() == weight.scalar_type(), "bias.scalar_type() == weight_type");
        TORCH_CHECK(bias_->is_cuda(), "bias.is_cuda()");
        TORCH_CHECK(bias_->stride(-1) == 1, "bias.stride(-1) == 1");
        TORCH_CHECK(bias_->dim() == 1 && bias_->size(0) == x.size(1), "bias must have shape (dim,)");
    }
}
void fill_conv(vms_conv_fwd_params& P, const Tensor& x, const Tensor& weight, const OptT& bias, const OptT& out, bool silu, bool reverse,
               int64_t reverse_from = 0) {
    P = vms_conv_fwd_params{};
    TORCH_CHECK(reverse_from >= 0 && reverse_from <= x.size(0) && (reverse_from == 0 || !reverse), "reverse_from must be in [0, batch] with reverse = False");
    P.reverse_from = (int)reverse_from;
    P.batch = (int)x.size(0); P.dim = (int)x.size(1); P.seqlen = (int)x.size(2); P.width = (int)weight.size(-1);
    P.dtype = dtype_code(x); P.wdtype = dtype_code(weight);
    P.silu_activation = silu; P.reverse = reverse;
    P.x = x.data_ptr(); P.weight = weight.data_ptr(); P.bias = cptr(bias); P.out = mptr(out);
    P.x_batch_stride = x.stride(0); P.x_c_stride = x.stride(1); P.x_l_stride = x.stride(2);
    P.weight_c_stride = weight.stride(0); P.weight_width_stride = weight.stride(1);
    if (out.has_value()) { P.out_batch_stride = out->stride(0); P.out_c_stride = out->stride(1); P.out_l_stride = out->stride(2); }
}
Tensor channel_last_like(const Tensor& x) {
    return at::empty({x.size(0), x.size(2), x.size(1)}, x.options()).transpose(1, 2);
}

Tensor conv_fwd(const Tensor& x, const Tensor& weight, const OptT& bias_, bool silu, bool reverse, int64_t reverse_from = 0) {   // causal_conv1d.cpp:130-189
    TORCH_CHECK(x.dim() == 3, "x must be (batch, dim, seqlen)");
    conv_common(x, weight, bias_);
    TORCH_CHECK(x.stride(2) == 1 || x.stride(1) == 1, "x.stride(2) == 1 || x.stride(1) == 1");
    const bool channel_last = x.stride(1) == 1 && x.stride(2) > 1;
    if (channel_last) TORCH_CHECK(x.size(1) % 8 == 0, "causal_conv1d only supports channel dimension divisible by 8 for now");
    Tensor out = at::empty_like(x);
    if (channel_last && out.stride(1) != 1) out = channel_last_like(x);
    vms_conv_fwd_params P;
    fill_conv(P, x, weight, bias_, out, silu, reverse, reverse_from);
    call("vms_causal_conv1d_fwd", vms_causal_conv1d_fwd, P, x);
    return out;
}

// both directions of a bidirectional block in one pass over x (vms_hip.h vms_causal_conv1d_fwd_dual): -> {conv(x; weight, bias),
// the anti-causal conv(x; weight_b, bias_b)}, both (batch, dim, seqlen) in physical order
std::vector<Tensor> conv_fwd_dual(const Tensor& x, const Tensor& weight, const OptT& bias_, const Tensor& weight_b, const OptT& bias_b_, bool silu) {
    TORCH_CHECK(x.dim() == 3 && x.stride(2) == 1, "x must be (batch, dim, seqlen) with a unit seqlen stride");
    conv_common(x, weight, bias_);
    conv_common(x, weight_b, bias_b_);
    TORCH_CHECK(weight.sizes() == weight_b.sizes() && weight.scalar_type() == weight_b.scalar_type() && bias_.has_value() == bias_b_.has_value(),
                "the two filters need the same shape, dtype and bias presence");
    Tensor out = at::empty(x.sizes(), x.options()), out_b = at::empty(x.sizes(), x.options());
    vms_conv_fwd_dual_params Q{};
    fill_conv(Q.f, x, weight, bias_, out, silu, false, 0);
    Q.weight_b = weight_b.data_ptr(); Q.bias_b = cptr(bias_b_); Q.out_b = out_b.data_ptr();
    Q.weight_b_c_stride = weight_b.stride(0); Q.weight_b_width_stride = weight_b.stride(1);
    Q.out_b_batch_stride = out_b.stride(0); Q.out_b_c_stride = out_b.stride(1);
    call("vms_causal_conv1d_fwd_dual", vms_causal_conv1d_fwd_dual, Q, x);
    return {out, out_b};
}

// The head of a bidirectional block's forward in one pass over x (vms_hip.h vms_conv_xproj_dual):
// -> {conv1d_out, conv1d_out_b, x_dbl, x_dbl_b}, or an empty vector when the kernel does not take the problem (the caller then runs
// conv_fwd_dual + x_proj_dual)
std::vector<Tensor> conv_xproj_dual(const Tensor& x, const Tensor& weight, const OptT& bias_, const Tensor& weight_b, const OptT& bias_b_,
                                    const Tensor& w_x, const Tensor& w_x_b) {
    const auto t16 = [](const Tensor& t) { return t.scalar_type() == at::kBFloat16 || t.scalar_type() == at::kHalf; };
    if (!(x.is_cuda() && x.dim() == 3 && x.stride(2) == 1 && t16(x) && w_x.scalar_type() == x.scalar_type() && w_x_b.scalar_type() == x.scalar_type() &&
          (weight.scalar_type() == at::kFloat || weight.scalar_type() == x.scalar_type()) && weight_b.scalar_type() == weight.scalar_type() &&
          weight.dim() == 2 && weight.sizes() == weight_b.sizes() && weight.size(1) >= 2 && weight.size(1) <= 4 && bias_.has_value() == bias_b_.has_value()))
        return {};
    if (bias_.has_value() && (bias_->scalar_type() != weight.scalar_type() || bias_b_->scalar_type() != weight.scalar_type())) return {};
    const int64_t b = x.size(0), d = x.size(1), L = x.size(2), m = w_x.size(0);
    if (!(w_x.dim() == 2 && w_x.size(1) == d && w_x.sizes() == w_x_b.sizes() && w_x.strides() == w_x_b.strides() && m >= 1 && m <= 96 &&
          w_x.stride(1) == 1 && w_x.stride(0) % 8 == 0 && weight.size(0) == d))
        return {};
    const auto al16 = [](const Tensor& t) { return (reinterpret_cast<uintptr_t>(t.data_ptr()) & 15) == 0; };
    if (L % 8 || d % 8 || x.stride(0) % 8 || x.stride(1) % 8 || x.stride(1) < L || !al16(x) || !al16(w_x) || !al16(w_x_b)) return {};
    if (((d - 1) * x.stride(1) + L) * 2 >= ((int64_t)1 << 31) || d * L * 2 >= ((int64_t)1 << 31)) return {};
    c10::DeviceGuard guard(x.device());
    const bool cslow = L <= 16 && b * d >= 4096 && b > 1 && x.stride(0) == L;      // (see kmajor below: conv1d_out channel-slowest like x)
    Tensor out = cslow ? at::empty({d, b, L}, x.options()).permute({1, 0, 2}) : at::empty(x.sizes(), x.options());
    Tensor out_b = cslow ? at::empty({d, b, L}, x.options()).permute({1, 0, 2}) : at::empty(x.sizes(), x.options());
    // short sequences with many rows (the lane-per-row scans' shapes) and a channel-slowest x: x_dbl row-major over (batch, position)
    // too, so that the products over it see batch x seqlen positions as one run (vms_proj_apply / vms_proj_wgrad fold such operands)
    const bool kmajor = L <= 16 && b * d >= 4096 && b > 1 && x.stride(0) == L;
    Tensor xa = kmajor ? at::empty({m, b, L}, x.options()).permute({1, 0, 2}) : at::empty({b, m, L}, x.options());
    Tensor xb = kmajor ? at::empty({m, b, L}, x.options()).permute({1, 0, 2}) : at::empty({b, m, L}, x.options());
    vms_conv_xproj_dual_params Q{};
    fill_conv(Q.c.f, x, weight, bias_, out, true, false, 0);
    Q.c.weight_b = weight_b.data_ptr(); Q.c.bias_b = cptr(bias_b_); Q.c.out_b = out_b.data_ptr();
    Q.c.weight_b_c_stride = weight_b.stride(0); Q.c.weight_b_width_stride = weight_b.stride(1);
    Q.c.out_b_batch_stride = out_b.stride(0); Q.c.out_b_c_stride = out_b.stride(1);
    Q.w_x = w_x.data_ptr(); Q.w_x_b = w_x_b.data_ptr(); Q.x_dbl = xa.data_ptr(); Q.x_dbl_b = xb.data_ptr();
    Q.m = (int)m; Q.tile = 0;
    Q.wx_row_stride = w_x.stride(0);
    Q.xdbl_batch_stride = xa.stride(0); Q.xdbl_row_stride = xa.stride(1);
    call("vms_conv_xproj_dual", vms_conv_xproj_dual, Q, x);
    return {out, out_b, xa, xb};
}

std::vector<OptT> conv_bwd(const Tensor& x, const Tensor& weight, const OptT& bias_, Tensor dout, const OptT& dx_, bool silu,
                           bool reverse, const OptT& zeroed, bool accumulate_dx, int64_t reverse_from = 0) {   // causal_conv1d.cpp:191-268
    TORCH_CHECK(x.dim() == 3, "x must be (batch, dim, seqlen)");
    conv_common(x, weight, bias_);
    TORCH_CHECK(dout.is_cuda(), "dout.is_cuda()");
    TORCH_CHECK(dout.sizes() == x.sizes(), "dout must have the shape of x");
    TORCH_CHECK(x.stride(2) == 1 || x.stride(1) == 1, "x.stride(2) == 1 || x.stride(1) == 1");
    const bool channel_last = x.stride(1) == 1 && x.stride(2) > 1;
    if (!channel_last && dout.stride(2) != 1) dout = dout.contiguous();
    if (channel_last && dout.stride(1) != 1) dout = dout.transpose(-1, -2).contiguous().transpose(-1, -2);
    Tensor dx;
    if (dx_.has_value()) {
        dx = *dx_;
        TORCH_CHECK(dx.scalar_type() == x.scalar_type(), "dx.scalar_type() == input_type");
        TORCH_CHECK(dx.is_cuda(), "dx.is_cuda()");
        TORCH_CHECK(dx.sizes() == x.sizes(), "dx must have the shape of x");
        TORCH_CHECK(channel_last ? dx.stride(1) == 1 : dx.stride(2) == 1, "dx must have x's unit-stride axis");
    } else {
        dx = at::empty_like(x);
        if (channel_last && dx.stride(1) != 1) dx = channel_last_like(x);
    }
    Tensor dweight;
    OptT dbias;
    if (zeroed.has_value()) {
        const int64_t nw = weight.numel(), nb = bias_.has_value() ? bias_->numel() : 0;
        TORCH_CHECK(zeroed->scalar_type() == at::kFloat && zeroed->is_cuda() && zeroed->dim() == 1 && zeroed->is_contiguous() && zeroed->numel() >= nw + nb,
                    "zeroed must be a flat float32 tensor of weight.numel() + bias.numel() elements");
        dweight = zeroed->narrow(0, 0, nw).view(weight.sizes());
        if (bias_.has_value()) dbias = zeroed->narrow(0, nw, nb);
    } else {
        dweight = at::zeros(weight.sizes(), weight.options().dtype(at::kFloat));
        if (bias_.has_value()) dbias = at::zeros(bias_->sizes(), bias_->options().dtype(at::kFloat));
    }
    TORCH_CHECK(!accumulate_dx || dx_.has_value(), "accumulate_dx needs the dx tensor to add to");
    vms_conv_bwd_params Q{};
    fill_conv(Q.f, x, weight, bias_, c10::nullopt, silu, reverse, reverse_from);
    Q.dout = dout.data_ptr(); Q.dx = dx.data_ptr(); Q.dweight = dweight.data_ptr<float>();
    Q.dbias = dbias.has_value() ? dbias->data_ptr<float>() : nullptr;
    Q.dout_batch_stride = dout.stride(0); Q.dout_c_stride = dout.stride(1); Q.dout_l_stride = dout.stride(2);
    Q.dx_batch_stride = dx.stride(0); Q.dx_c_stride = dx.stride(1); Q.dx_l_stride = dx.stride(2);
    Q.dweight_c_stride = dweight.stride(0); Q.dweight_width_stride = dweight.stride(1);
    Q.dx_accumulate = accumulate_dx;
    call("vms_causal_conv1d_bwd", vms_causal_conv1d_bwd, Q, x);
    OptT db;
    if (bias_.has_value()) db = dbias->to(bias_->scalar_type());
    return {dx, dweight.to(weight.scalar_type()), db};
}

Tensor conv_update(const Tensor& x, const Tensor& conv_state, const Tensor& weight, const OptT& bias_, bool silu) {   // causal_conv1d.cpp:270-327
    TORCH_CHECK(x.dim() == 2, "x must be (batch, dim)");
    conv_common(x, weight, bias_);
    TORCH_CHECK(conv_state.scalar_type() == x.scalar_type(), "conv_state.scalar_type() == input_type");
    TORCH_CHECK(conv_state.is_cuda(), "conv_state.is_cuda()");
    TORCH_CHECK(conv_state.dim() == 3 && conv_state.size(0) == x.size(0) && conv_state.size(1) == x.size(1) && conv_state.size(2) == weight.size(-1),
                "conv_state must have shape (batch, dim, width)");
    Tensor out = at::empty_like(x);
    vms_conv_fwd_params P{};
    P.batch = (int)x.size(0); P.dim = (int)x.size(1); P.seqlen = 1; P.width = (int)weight.size(-1);
    P.dtype = dtype_code(x); P.wdtype = dtype_code(weight); P.silu_activation = silu;
    P.x = x.data_ptr(); P.weight = weight.data_ptr(); P.bias = cptr(bias_); P.out = out.data_ptr();
    P.x_batch_stride = x.stride(0); P.x_c_stride = x.stride(1); P.x_l_stride = 1;
    P.weight_c_stride = weight.stride(0); P.weight_width_stride = weight.stride(1);
    P.out_batch_stride = out.stride(0); P.out_c_stride = out.stride(1); P.out_l_stride = 1;
    P.conv_state = conv_state.data_ptr();
    P.conv_state_batch_stride = conv_state.stride(0); P.conv_state_c_stride = conv_state.stride(1); P.conv_state_l_stride = conv_state.stride(2);
    call("vms_causal_conv1d_update", vms_causal_conv1d_update, P, x);
    return out;
}


// ---- the fused inner node of the Mamba block in one host call -----------------------------------------------------------
// conv1d + SiLU -> x_proj GEMM -> dt_proj GEMM -> selective scan (+ z gate), and its backward, for the case every module of
// the suite runs: input-dependent B / C, real A, no fused out_proj, no B / C projection biases
// (mamba_ssm/ops/selective_scan_interface.py _inner_forward / _inner_backward, reference SSI:155-289).  Same operations in
// the same order as the Python statement of the node -- the small GEMMs through ATen (hipBLASLt), the kernels through the
// C ABI -- but one Python -> C++ crossing per direction instead of ~25: launch-bound shapes (the DBM block at
// (2, 2304, 512)) are bound by exactly that host time.
struct PaddedBC { Tensor B, C; int64_t pad; };
PaddedBC pad_bc(const Tensor& B, const Tensor& C, bool reverse, bool both = false) {   // selective_scan_cuda.pad_bc
    const int64_t L = B.size(-1), pad = (16 - L % 16) % 16;
    if (pad == 0) return {B, C, 0};
    if (both) return {at::constant_pad_nd(B, {pad, pad}).narrow(-1, pad, L), at::constant_pad_nd(C, {pad, pad}).narrow(-1, pad, L), pad};
    if (reverse) return {at::constant_pad_nd(B, {pad, 0}).narrow(-1, pad, L), at::constant_pad_nd(C, {pad, 0}).narrow(-1, pad, L), pad};
    return {at::constant_pad_nd(B, {0, pad}).narrow(-1, 0, L), at::constant_pad_nd(C, {0, pad}).narrow(-1, 0, L), pad};
}

// ---- the inner node's small projections on the matrix cores (vms_hip.h vms_proj_apply / vms_proj_wgrad) ----------------
// Eligible: 16-bit tensors of one dtype, unit seqlen strides, seqlen and the row / batch strides multiples of 8 elements,
// 16-byte aligned bases, k <= 96 (apply) / m <= 128 (wgrad).  Anything else stays with the library GEMM the caller had.
bool proj_ok16(const Tensor& t) {
    return t.is_cuda() && (t.scalar_type() == at::kBFloat16 || t.scalar_type() == at::kHalf) && t.dim() == 3 && t.stride(2) == 1 &&
           t.size(2) % 8 == 0 && t.stride(0) % 8 == 0 && t.stride(1) % 8 == 0 && (reinterpret_cast<uintptr_t>(t.data_ptr()) & 15) == 0;
}
bool proj_apply_eligible(const Tensor& w, const Tensor& in, const Tensor& out) {
    const auto fits = [](const Tensor& t) { return ((t.size(1) - 1) * t.stride(1) + t.size(2)) * 2 < ((int64_t)1 << 31); };   // one buffer resource per batch entry
    return proj_ok16(in) && proj_ok16(out) && w.is_cuda() && w.dim() == 2 && w.scalar_type() == in.scalar_type() &&
           out.scalar_type() == in.scalar_type() && w.size(1) == in.size(1) && w.size(1) <= 96 && out.size(1) == w.size(0) &&
           out.size(0) == in.size(0) && out.size(2) == in.size(2) && fits(in) && fits(out);
}
// out[b, d, l] (+)= sum_r w[d, r] in[b, r, l]
void proj_apply(const Tensor& w, const Tensor& in, const Tensor& out, bool accumulate) {
    vms_proj_apply_params P{};
    P.batch = (int)in.size(0); P.rows = (int)w.size(0); P.k = (int)w.size(1); P.seqlen = (int)in.size(2);
    P.dtype = dtype_code(in); P.accumulate = accumulate;
    P.w = w.data_ptr(); P.in = in.data_ptr(); P.out = out.data_ptr();
    P.w_row_stride = w.stride(0); P.w_k_stride = w.stride(1);
    P.in_batch_stride = in.stride(0); P.in_k_stride = in.stride(1);
    P.out_batch_stride = out.stride(0); P.out_row_stride = out.stride(1);
    call("vms_proj_apply", vms_proj_apply, P, in);
}
bool proj_wgrad_eligible(const Tensor& p, const Tensor& q) {
    const auto fits = [](const Tensor& t) { return ((t.size(1) - 1) * t.stride(1) + t.size(2)) * 2 < ((int64_t)1 << 31); };   // one buffer resource per batch entry
    return proj_ok16(p) && proj_ok16(q) && p.scalar_type() == q.scalar_type() && p.size(0) == q.size(0) && p.size(2) == q.size(2) &&
           p.size(1) <= 128 && fits(p) && fits(q);
}
// dw[m, n] += sum_{b, l} p[b, m, l] q[b, n, l];  dw: fp32 (m, n), unit column stride, zero-filled by the caller
// transposed: dw is the (n, m) matrix (a parameter stored that way receives a gradient in its own layout)
void proj_wgrad(const Tensor& p, const Tensor& q, const Tensor& dw, bool transposed = false) {
    TORCH_CHECK(dw.scalar_type() == at::kFloat && dw.dim() == 2 && dw.stride(1) == 1 && dw.size(transposed ? 1 : 0) == p.size(1) &&
                    dw.size(transposed ? 0 : 1) == q.size(1),
                "proj_wgrad: dw must be a float32 (m, n) matrix -- (n, m) when transposed -- with unit column stride");
    vms_proj_wgrad_params P{};
    P.batch = (int)p.size(0); P.m = (int)p.size(1); P.n = (int)q.size(1); P.seqlen = (int)p.size(2);
    P.dtype = dtype_code(p); P.dw_transposed = transposed;
    P.p = p.data_ptr(); P.q = q.data_ptr(); P.dw = dw.data_ptr<float>();
    P.p_batch_stride = p.stride(0); P.p_row_stride = p.stride(1);
    P.q_batch_stride = q.stride(0); P.q_row_stride = q.stride(1);
    P.dw_row_stride = dw.stride(0);
    call("vms_proj_wgrad", vms_proj_wgrad, P, p);
}

// out[b, m, l] = sum_k w[m, k] in[b, k, l]  (vms_hip.h vms_proj_kred): x_dbl = x_proj_w @ conv_out, dx_dbl[:R] = dt_proj_w^T @ ddelta
bool proj_kred_eligible(const Tensor& w, const Tensor& in, const Tensor& out) {
    if (!(proj_ok16(in) && out.is_cuda() && out.dim() == 3 && out.stride(2) == 1 && out.scalar_type() == in.scalar_type() && w.is_cuda() &&
          w.dim() == 2 && w.scalar_type() == in.scalar_type()))
        return false;
    const int64_t m = w.size(0), k = w.size(1);
    if (!(m >= 1 && m <= 96 && in.size(1) == k && out.size(0) == in.size(0) && out.size(1) == m && out.size(2) == in.size(2))) return false;
    if (reinterpret_cast<uintptr_t>(w.data_ptr()) & 15) return false;
    if (((k - 1) * in.stride(1) + in.size(2)) * 2 >= ((int64_t)1 << 31)) return false;   // a batch entry is addressed through one buffer resource
    return (w.stride(1) == 1 && w.stride(0) % 8 == 0 && k % 8 == 0) || (w.stride(0) == 1 && w.stride(1) % 8 == 0 && m % 8 == 0);
}
// cast_src: fp32 (groups, batch, rows, seqlen), rounded into the groups * rows rows that follow out's m rows in its parent tensor
void proj_kred(const Tensor& w, const Tensor& in, const Tensor& out, const OptT& w2, const OptT& in2, const OptT& out2, int64_t tile,
               const OptT& cast_src = OptT(), const OptT& cast_src2 = OptT()) {
    TORCH_CHECK(proj_kred_eligible(w, in, out), "proj_kred: 16-bit w (m <= 96, k), in (batch, k, seqlen), out (batch, m, seqlen) of one dtype "
                "expected; unit seqlen strides, seqlen / strides multiples of 8, 16-byte aligned, w contiguous along k or m");
    vms_proj_kred_params P{};
    P.batch = (int)in.size(0); P.m = (int)w.size(0); P.k = (int)w.size(1); P.seqlen = (int)in.size(2);
    P.dtype = dtype_code(in); P.tile = (int)tile;
    P.w = w.data_ptr(); P.in = in.data_ptr(); P.out = out.data_ptr();
    P.w_row_stride = w.stride(0); P.w_k_stride = w.stride(1);
    P.in_batch_stride = in.stride(0); P.in_k_stride = in.stride(1);
    P.out_batch_stride = out.stride(0); P.out_row_stride = out.stride(1);
    if (w2.has_value()) {
        TORCH_CHECK(in2.has_value() && out2.has_value() && proj_kred_eligible(*w2, *in2, *out2) && w2->sizes() == w.sizes() &&
                    in2->sizes() == in.sizes() && w2->strides() == w.strides() && in2->strides() == in.strides() && out2->strides() == out.strides(),
                    "proj_kred: the second problem must have the first one's shapes, strides and dtype");
        P.w2 = w2->data_ptr(); P.in2 = in2->data_ptr(); P.out2 = out2->data_ptr();
    }
    if (cast_src.has_value()) {
        const Tensor& c = *cast_src;
        TORCH_CHECK(c.scalar_type() == at::kFloat && c.dim() == 4 && c.stride(3) == 1 && c.size(1) == in.size(0) && c.size(3) == in.size(2),
                    "proj_kred: cast_src must be fp32 (groups, batch, rows, seqlen) with a unit seqlen stride");
        P.cast_src = c.data_ptr<float>(); P.cast_groups = (int)c.size(0); P.cast_rows = (int)c.size(2);
        P.cast_group_stride = c.stride(0); P.cast_batch_stride = c.stride(1); P.cast_row_stride = c.stride(2);
        if (cast_src2.has_value()) {
            TORCH_CHECK(w2.has_value() && cast_src2->scalar_type() == at::kFloat && cast_src2->sizes() == c.sizes() && cast_src2->strides() == c.strides(),
                        "proj_kred: cast_src2 comes with the second problem and has cast_src's shape and strides");
            P.cast_src2 = cast_src2->data_ptr<float>();
        }
    }
    call("vms_proj_kred", vms_proj_kred, P, in);
}
// x_dbl of both directions of a bidirectional block as one launch (or two library GEMMs when the kernel declines)
std::vector<Tensor> x_proj_dual(const Tensor& w_a, const Tensor& conv_out_a, const Tensor& w_b, const Tensor& conv_out_b, bool use_kred) {
    c10::DeviceGuard guard(conv_out_a.device());
    if (use_kred && conv_out_a.dim() == 3) {
        Tensor xa = at::empty({conv_out_a.size(0), w_a.size(0), conv_out_a.size(2)}, conv_out_a.options());
        Tensor xb = at::empty_like(xa);
        if (proj_kred_eligible(w_a, conv_out_a, xa) && proj_kred_eligible(w_b, conv_out_b, xb) && w_a.sizes() == w_b.sizes() &&
            w_a.strides() == w_b.strides() && conv_out_a.sizes() == conv_out_b.sizes() && conv_out_a.strides() == conv_out_b.strides()) {
            proj_kred(w_a, conv_out_a, xa, w_b, conv_out_b, xb, 0);
            return {xa, xb};
        }
    }
    return {at::matmul(w_a, conv_out_a), at::matmul(w_b, conv_out_b)};
}

bool proj_conv_bwd_eligible(const Tensor& x, const Tensor& du_like, const Tensor& dx_dbl_like, const Tensor& w_x, const Tensor& conv_w,
                            const OptT& conv_b, const Tensor& dx) {
    const int64_t k = dx_dbl_like.size(1);
    // seqlen % 8 == 0: whole 16-byte pieces (aligned bases / strides); otherwise the ragged flavour, any 2-byte alignment
    auto ok = [&](const Tensor& t) {
        if (x.size(2) % 8 == 0) return proj_ok16(t);
        return t.is_cuda() && (t.scalar_type() == at::kBFloat16 || t.scalar_type() == at::kHalf) && t.dim() == 3 && t.stride(2) == 1 &&
               t.stride(1) >= t.size(2);
    };
    // a batch entry of each tensor is addressed through one buffer resource: (rows - 1) * row stride + seqlen elements must span < 2 GiB
    auto fits = [&](const Tensor& t) { return ((t.size(1) - 1) * t.stride(1) + t.size(2)) * 2 < ((int64_t)1 << 31) && t.size(1) * t.stride(1) * 2 < ((int64_t)1 << 31); };
    return ok(x) && ok(du_like) && ok(dx_dbl_like) && ok(dx) && fits(x) && fits(du_like) && fits(dx_dbl_like) && fits(dx) &&
           w_x.scalar_type() == x.scalar_type() &&
           du_like.scalar_type() == x.scalar_type() && dx_dbl_like.scalar_type() == x.scalar_type() && dx.scalar_type() == x.scalar_type() &&
           k >= 1 && k <= 96 && conv_w.dim() == 2 && conv_w.size(1) >= 2 && conv_w.size(1) <= 4 && is_itype(conv_w) &&
           (!conv_b.has_value() || conv_b->scalar_type() == conv_w.scalar_type());
}
// vms_hip.h vms_proj_conv_bwd: dw_x += dx_dbl conv1d_out^T; dx, dconv_w, dconv_b = conv1d backward of (du + w_x^T dx_dbl)
void proj_conv_bwd(const Tensor& x, const Tensor& du, const Tensor& dx_dbl, const Tensor& w_x, const Tensor& conv_w, const OptT& conv_b,
                   const Tensor& dx, const Tensor& dconv_w, const OptT& dconv_b, const Tensor& dw_x, bool reverse, int64_t reverse_from,
                   bool accumulate_dx) {
    vms_proj_conv_bwd_params P{};
    P.batch = (int)x.size(0); P.dim = (int)x.size(1); P.seqlen = (int)x.size(2); P.k = (int)dx_dbl.size(1); P.width = (int)conv_w.size(1);
    P.dtype = dtype_code(x); P.wdtype = dtype_code(conv_w);
    P.reverse = reverse; P.reverse_from = (int)reverse_from; P.dx_accumulate = accumulate_dx;
    P.x = x.data_ptr(); P.du = du.data_ptr(); P.dx_dbl = dx_dbl.data_ptr(); P.w_x = w_x.data_ptr();
    P.conv_weight = conv_w.data_ptr(); P.conv_bias = cptr(conv_b);
    P.dx = dx.data_ptr(); P.dconv_weight = dconv_w.data_ptr<float>(); P.dconv_bias = dconv_b.has_value() ? dconv_b->data_ptr<float>() : nullptr;
    P.dw_x = dw_x.data_ptr<float>();
    P.x_batch_stride = x.stride(0); P.x_c_stride = x.stride(1);
    P.du_batch_stride = du.stride(0); P.du_c_stride = du.stride(1);
    P.dxdbl_batch_stride = dx_dbl.stride(0); P.dxdbl_k_stride = dx_dbl.stride(1);
    P.wx_k_stride = w_x.stride(0); P.wx_c_stride = w_x.stride(1);
    P.conv_weight_c_stride = conv_w.stride(0); P.conv_weight_width_stride = conv_w.stride(1);
    P.dx_batch_stride = dx.stride(0); P.dx_c_stride = dx.stride(1);
    P.dconv_weight_c_stride = dconv_w.stride(0); P.dconv_weight_width_stride = dconv_w.stride(1);
    P.dwx_k_stride = dw_x.stride(0);
    call("vms_proj_conv_bwd", vms_proj_conv_bwd, P, x);
}

// -> [out_z, conv_out, x_dbl, delta, ckpt, out]
std::vector<Tensor> inner_fwd(const Tensor& xz, const Tensor& conv_w, const OptT& conv_b, const Tensor& x_proj_w, const Tensor& dt_proj_w,
                              const Tensor& A, const OptT& D_, const OptT& delta_bias_, bool delta_softplus, bool reverse,
                              const OptT& out_z_into, int64_t impl, int64_t segments, int64_t reverse_from, int64_t proj_flags,
                              const OptT& conv_out_given, const OptT& x_dbl_given, int64_t seq_valid) {
    TORCH_CHECK(xz.is_cuda() && xz.dim() == 3 && xz.stride(2) == 1, "xz must be a (batch, 2 * dim, seqlen) GPU tensor with unit seqlen stride");
    TORCH_CHECK(seq_valid >= 0 && seq_valid <= xz.size(2) && (seq_valid == 0 || delta_softplus), "inner_fwd: 0 <= seq_valid <= seqlen, and only with delta_softplus");
    c10::DeviceGuard guard(xz.device());
    const int64_t d = conv_w.size(0), R = dt_proj_w.size(1), N = A.size(1);
    TORCH_CHECK(xz.size(1) == 2 * d && x_proj_w.size(0) == R + 2 * N && x_proj_w.size(1) == d && dt_proj_w.size(0) == d,
                "inner_fwd: xz (b, 2d, l), x_proj (R + 2N, d), dt_proj (d, R), A (d, N) expected");
    const Tensor x = xz.narrow(1, 0, d), z = xz.narrow(1, d, d);
    // conv_out_given: this direction's conv1d output, already computed (both directions of a block by one conv_fwd_dual)
    Tensor conv_out = conv_out_given.has_value() ? *conv_out_given : conv_fwd(x, conv_w, conv_b, true, reverse, reverse_from);
    // x_dbl_given: x_proj_w @ conv_out, already computed (right behind the conv1d that wrote conv_out, while it is in the Infinity Cache)
    Tensor x_dbl;                                                        // (b, R + 2N, l): rows R.. are B, the last N are C
    if (x_dbl_given.has_value()) {
        x_dbl = *x_dbl_given;
    } else {
        x_dbl = at::empty({conv_out.size(0), R + 2 * N, conv_out.size(2)}, conv_out.options());
        if ((proj_flags & 16) && proj_kred_eligible(x_proj_w, conv_out, x_dbl)) proj_kred(x_proj_w, conv_out, x_dbl, OptT(), OptT(), OptT(), 0);
        else at::matmul_out(x_dbl, x_proj_w, conv_out);
    }
    Tensor delta;                                                        // (b, d, l) = dt_proj_w @ x_dbl[:, :R]
    {
        const Tensor dt_in = x_dbl.narrow(1, 0, R);
        // (x_dbl row-major over (batch, position) -- conv_xproj_dual's layout for short sequences: delta channel-slowest likewise)
        const bool kmajor = x_dbl.size(0) > 1 && x_dbl.stride(0) == x_dbl.size(2) && x_dbl.stride(2) == 1;
        delta = kmajor ? at::empty({d, x_dbl.size(0), x_dbl.size(2)}, x_dbl.options()).permute({1, 0, 2})
                       : at::empty({x_dbl.size(0), d, x_dbl.size(2)}, x_dbl.options());
        if ((proj_flags & 1) && proj_apply_eligible(dt_proj_w, dt_in, delta)) proj_apply(dt_proj_w, dt_in, delta, false);
        else at::matmul_out(delta, dt_proj_w, dt_in);
        // seq_valid: the positions behind it are the mixer's zero padding to whole vectors (modules/_core.py): softplus(-inf) = 0 makes
        // them identity steps of the recurrence (a = 1, b = 0) in either direction, and every gradient through them exactly 0
        if (seq_valid > 0 && seq_valid < delta.size(2))
            delta.narrow(2, seq_valid, delta.size(2) - seq_valid).fill_(-std::numeric_limits<float>::infinity());
    }
    const PaddedBC bc = pad_bc(x_dbl.narrow(1, R, N).unsqueeze(1), x_dbl.narrow(1, R + N, N).unsqueeze(1), reverse, reverse_from > 0);
    // proj_flags bit 4: nothing will run this node's backward (small checkpoint layout); bit 8: keep the 128-element checkpoints
    std::vector<Tensor> r = scan_fwd(conv_out, delta, A, bc.B, bc.C, D_, z, delta_bias_, delta_softplus, reverse, out_z_into, bc.pad, impl, segments,
                                     reverse_from, (proj_flags & 12) ? 1 : -1);
    return {r[2], conv_out, x_dbl, delta, r[1], r[0]};
}

// The node's backward around its scan: what is decided and allocated before the scan (begin) and everything after it (finish).
struct InnerBwd {
    Tensor xz, x, z, dxz, dx, dz, Bv, Cv, dt_in, zeros, zeros_spare, conv_w, x_proj_w, dt_proj_w, conv_out, x_dbl, A;
    OptT conv_b, D_, delta_bias_;
    PaddedBC bc;
    int64_t b, d, R, N, K2, n_scan, n_conv, n_proj;
    bool acc, use_mfma_proj, use_kred, mfma_wg, fused_tail, reverse;
    int64_t reverse_from;
    at::ScalarType wdt;
    Tensor dx_dbl_pre;      // set by inner_bwd_dual: dx_dbl with its d_dt and dB / dC rows already written (one vms_proj_kred for both directions)
};

InnerBwd inner_bwd_begin(const Tensor& xz, const Tensor& conv_w, const OptT& conv_b, const Tensor& x_proj_w, const Tensor& dt_proj_w,
                         const Tensor& A, const OptT& D_, const OptT& delta_bias_, const Tensor& conv_out, const Tensor& x_dbl,
                         const Tensor& delta, bool reverse, const OptT& dxz_into, int64_t reverse_from, bool wgrad_fp32, int64_t proj_flags,
                         bool zeros_for_two = false, const OptT& zeros_given = OptT()) {
    InnerBwd I;
    I.use_mfma_proj = (proj_flags & 1) != 0;
    I.use_kred = (proj_flags & 16) != 0;
    I.wdt = wgrad_fp32 ? at::kFloat : x_proj_w.scalar_type();   // the parameters' dtype: autograd has nothing to cast
    I.b = xz.size(0); I.d = conv_w.size(0); I.R = dt_proj_w.size(1); I.N = A.size(1);
    I.xz = xz; I.conv_w = conv_w; I.conv_b = conv_b; I.x_proj_w = x_proj_w; I.dt_proj_w = dt_proj_w; I.conv_out = conv_out; I.x_dbl = x_dbl;
    I.A = A; I.D_ = D_; I.delta_bias_ = delta_bias_; I.reverse = reverse; I.reverse_from = reverse_from;
    I.x = xz.narrow(1, 0, I.d); I.z = xz.narrow(1, I.d, I.d);
    I.acc = dxz_into.has_value();
    I.dxz = I.acc ? *dxz_into : at::empty_like(xz);
    I.dx = I.dxz.narrow(1, 0, I.d); I.dz = I.dxz.narrow(1, I.d, I.d);
    I.Bv = x_dbl.narrow(1, I.R, I.N).unsqueeze(1); I.Cv = x_dbl.narrow(1, I.R + I.N, I.N).unsqueeze(1);
    // one zero fill for every fp32 atomics target of the node (scan: dA, dB, dC, dD, ddelta_bias; conv: dweight, dbias)
    I.n_scan = A.numel() + 2 * I.Bv.numel() + (D_.has_value() ? D_->numel() : 0) + (delta_bias_.has_value() ? delta_bias_->numel() : 0);
    I.n_conv = conv_w.numel() + (conv_b.has_value() ? conv_b->numel() : 0);
    // the two small weight gradients on the matrix cores accumulate in fp32 into the same zero-filled buffer
    I.dt_in = x_dbl.narrow(1, 0, I.R);
    I.mfma_wg = I.use_mfma_proj && proj_wgrad_eligible(x_dbl, conv_out) && proj_wgrad_eligible(I.dt_in, delta);
    I.n_proj = I.mfma_wg ? (I.R + (I.R + 2 * I.N)) * I.d : 0;
    I.K2 = I.R + 2 * I.N;
    I.fused_tail = (proj_flags & 2) && proj_conv_bwd_eligible(I.x, conv_out, x_dbl, x_proj_w, conv_w, conv_b, I.dx);
    // zeros_for_two: ONE fill for both directions of a bidirectional block (the second direction's begin takes `zeros_spare`)
    const int64_t n_zero = I.n_scan + I.n_conv + I.n_proj + (I.fused_tail ? I.K2 * I.d : 0), n_pad = (n_zero + 63) / 64 * 64;   // 256-byte slices
    if (zeros_given.has_value() && zeros_given->numel() >= n_zero) {
        I.zeros = zeros_given->narrow(0, 0, n_zero);
    } else if (zeros_for_two) {
        const Tensor both = at::zeros({2 * n_pad}, xz.options().dtype(at::kFloat));
        I.zeros = both.narrow(0, 0, n_zero);
        I.zeros_spare = both.narrow(0, n_pad, n_pad);
    } else {
        I.zeros = at::zeros({n_zero}, xz.options().dtype(at::kFloat));
    }
    I.bc = pad_bc(I.Bv, I.Cv, reverse, reverse_from > 0);
    return I;
}

// g = the scan's [du, ddelta, dA, dB, dC, dD, ddelta_bias, ...]; dB / dC were accumulated in I.zeros
std::vector<OptT> inner_bwd_finish(InnerBwd& I, const std::vector<OptT>& g) {
    const int64_t b = I.b, d = I.d, R = I.R, N = I.N, K2 = I.K2, n_scan = I.n_scan, n_conv = I.n_conv, n_proj = I.n_proj;
    const Tensor &x_dbl = I.x_dbl, &conv_out = I.conv_out, &x_proj_w = I.x_proj_w, &dt_proj_w = I.dt_proj_w, &conv_w = I.conv_w, &dt_in = I.dt_in;
    const OptT& conv_b = I.conv_b;
    Tensor& zeros = I.zeros;
    const auto wdt = I.wdt;
    Tensor dconv_out = *g[0], ddelta = *g[1];
    const bool pre = I.dx_dbl_pre.defined();
    Tensor dx_dbl = pre ? I.dx_dbl_pre : at::empty_like(x_dbl);                             // (b, R + 2N, l)
    // dB and dC sit back to back in the zero-filled buffer (scan_bwd carves dA, dB, dC, ...): rounded into their rows of dx_dbl by the
    // kernel that writes its first R rows (vms_proj_kred's cast rows), or by one cast kernel for both
    const Tensor dbc = zeros.narrow(0, I.A.numel(), 2 * I.Bv.numel()).view({2, b, N, dx_dbl.size(2)});
    Tensor d_dt = dx_dbl.narrow(1, 0, R);                                                   // (b, R, l) = W_dt^T ddelta
    const bool kred_dt = pre || (I.use_kred && proj_kred_eligible(dt_proj_w.t(), ddelta, d_dt) && dx_dbl.size(2) % 4 == 0);
    if (!kred_dt) dx_dbl.narrow(1, R, 2 * N).view({b, 2, N, dx_dbl.size(2)}).copy_(dbc.permute({1, 0, 2, 3}));
    Tensor ddt_proj_w, dx_proj_w;
    if (I.mfma_wg && proj_wgrad_eligible(dt_in, ddelta)) {
        Tensor dw1 = zeros.narrow(0, n_scan + n_conv, R * d).view({d, R});                  // (d, R): the parameter's own layout
        proj_wgrad(dt_in, ddelta, dw1, /*transposed=*/true);
        ddt_proj_w = dw1.to(wdt);
    } else {
        // (d, R) = sum over batch AND positions of ddelta dt_in^T.  With a small batch the library runs ONE skinny GEMM per entry
        // with K = seqlen (batch 1, 65,536 positions: 24 workgroups, 105 us): cut K into slices -- more, shorter GEMMs -- and
        // sum them with the batch
        const int64_t L = ddelta.size(2);
        int64_t ks = 1;
        while (b * ks < 8 && L % (2 * ks) == 0 && L / (2 * ks) >= 2048) ks *= 2;
        if (ks > 1 && ddelta.stride(2) == 1 && dt_in.stride(2) == 1) {
            const Tensor dd4 = ddelta.unflatten(2, {ks, L / ks}).permute({0, 2, 1, 3});          // (b, ks, d, L / ks)
            const Tensor di4 = dt_in.unflatten(2, {ks, L / ks}).permute({0, 2, 3, 1});           // (b, ks, L / ks, R)
            ddt_proj_w = at::sum(at::matmul(dd4, di4), at::IntArrayRef({0, 1}), false, wdt);
        } else {
            ddt_proj_w = at::sum(at::matmul(ddelta, dt_in.transpose(1, 2)), {0}, false, wdt);   // (d, R)
        }
    }
    // written straight into its rows of dx_dbl (a batch-strided output: no copy kernel)
    if (pre) {}
    else if (kred_dt) proj_kred(dt_proj_w.t(), ddelta, d_dt, OptT(), OptT(), OptT(), 0, dbc);
    else at::bmm_out(d_dt, dt_proj_w.t().unsqueeze(0).expand({b, -1, -1}), ddelta);
    if (I.fused_tail) {
        // SSI:278-283 in one pass over the activations (vms_proj_conv_bwd): dconv1d_out = du + W_x^T dx_dbl stays on chip
        Tensor dwx = zeros.narrow(0, n_scan + n_conv + n_proj, K2 * d).view({K2, d});
        Tensor dcw = zeros.narrow(0, n_scan, conv_w.numel()).view(conv_w.sizes());
        OptT dcb;
        if (conv_b.has_value()) dcb = zeros.narrow(0, n_scan + conv_w.numel(), conv_b->numel());
        proj_conv_bwd(I.x, dconv_out, dx_dbl, x_proj_w, conv_w, conv_b, I.dx, dcw, dcb, dwx, I.reverse, I.reverse_from, I.acc);
        OptT db;
        if (conv_b.has_value()) db = dcb->to(conv_b->scalar_type());
        return {I.dxz, dcw.to(conv_w.scalar_type()), db, dwx.to(wdt), ddt_proj_w, g[2], g[5], g[6]};
    }
    if (I.mfma_wg && proj_wgrad_eligible(dx_dbl, conv_out)) {
        Tensor dw2 = zeros.narrow(0, n_scan + n_conv + R * d, (R + 2 * N) * d).view({R + 2 * N, d});
        proj_wgrad(dx_dbl, conv_out, dw2);
        dx_proj_w = dw2.to(wdt);                                                            // (R + 2N, d)
    } else {
        dx_proj_w = at::sum(at::matmul(dx_dbl, conv_out.transpose(1, 2)), {0}, false, wdt);
    }
    if (I.use_mfma_proj && proj_apply_eligible(x_proj_w.t(), dx_dbl, dconv_out)) proj_apply(x_proj_w.t(), dx_dbl, dconv_out, true);
    else dconv_out.baddbmm_(x_proj_w.t().expand({b, -1, -1}), dx_dbl);                      // + W_x^T dx_dbl, in place
    std::vector<OptT> c = conv_bwd(I.x, conv_w, conv_b, dconv_out, I.dx, true, I.reverse, zeros.narrow(0, n_scan, n_conv), I.acc, I.reverse_from);
    return {I.dxz, c[1], c[2], dx_proj_w, ddt_proj_w, g[2], g[5], g[6]};
}

// -> [dxz, dconv_w (d, w), dconv_b | undefined, dx_proj_w, ddt_proj_w, dA, dD | undefined, ddelta_bias | undefined]
// dxz_into: a dxz that already holds what xz received through another node; this node's dx / dz are ADDED by the kernels.
std::vector<OptT> inner_bwd(const Tensor& dout_, const Tensor& xz, const Tensor& conv_w, const OptT& conv_b, const Tensor& x_proj_w,
                            const Tensor& dt_proj_w, const Tensor& A, const OptT& D_, const OptT& delta_bias_, const Tensor& conv_out,
                            const Tensor& x_dbl, const Tensor& delta, const Tensor& ckpt, const Tensor& out, bool delta_softplus,
                            bool reverse, const OptT& dxz_into, int64_t impl, int64_t segments, int64_t reverse_from, bool wgrad_fp32,
                            int64_t proj_flags) {
    c10::DeviceGuard guard(xz.device());
    const Tensor dout = dout_.stride(-1) == 1 ? dout_ : dout_.contiguous();
    InnerBwd I = inner_bwd_begin(xz, conv_w, conv_b, x_proj_w, dt_proj_w, A, D_, delta_bias_, conv_out, x_dbl, delta, reverse, dxz_into,
                                 reverse_from, wgrad_fp32, proj_flags);
    std::vector<OptT> g = scan_bwd(conv_out, delta, A, I.bc.B, I.bc.C, D_, I.z, delta_bias_, dout, ckpt, out, I.dz, delta_softplus,
                                   /*recompute_out_z=*/false, reverse, I.zeros.narrow(0, 0, I.n_scan), /*keep_fp32=*/true, I.acc, I.bc.pad, impl,
                                   segments, I.Bv, I.Cv, reverse_from);
    return inner_bwd_finish(I, g);
}

// Both directions of a bidirectional block (BiMambaInnerFnNoOutProj): `a` = the left-to-right parameter set, `b` = the
// right-to-left one, each given as [conv_w, conv_b, x_proj_w, dt_proj_w, A, D, delta_bias, conv_out, x_dbl, delta, ckpt, out]
// (conv_b / D / delta_bias may be None).  The two backward scans go out as ONE call (vms_selective_scan_bwd_dual: one grid when
// the pair qualifies); everything after the scans runs per direction as in inner_bwd, b's dx added to a's by its kernels.
// -> [dxz, then the 7 parameter gradients of a, then those of b]
std::vector<OptT> inner_bwd_dual(const Tensor& dout_, const Tensor& xz, const std::vector<OptT>& a, const std::vector<OptT>& b,
                                 bool delta_softplus, int64_t impl, int64_t segments, bool wgrad_fp32, int64_t proj_flags) {
    TORCH_CHECK(a.size() == 12 && b.size() == 12, "inner_bwd_dual: 12 tensors per direction expected");
    c10::DeviceGuard guard(xz.device());
    const Tensor dout = dout_.stride(-1) == 1 ? dout_ : dout_.contiguous();
    auto T = [](const OptT& t) -> const Tensor& { TORCH_CHECK(t.has_value(), "inner_bwd_dual: missing tensor"); return *t; };
    InnerBwd Ia = inner_bwd_begin(xz, T(a[0]), a[1], T(a[2]), T(a[3]), T(a[4]), a[5], a[6], T(a[7]), T(a[8]), T(a[9]), /*reverse=*/false,
                                  OptT(), 0, wgrad_fp32, proj_flags, /*zeros_for_two=*/true);
    InnerBwd Ib = inner_bwd_begin(xz, T(b[0]), b[1], T(b[2]), T(b[3]), T(b[4]), b[5], b[6], T(b[7]), T(b[8]), T(b[9]), /*reverse=*/true,
                                  Ia.dxz, 0, wgrad_fp32, proj_flags, false, Ia.zeros_spare);
    std::vector<std::vector<OptT>> g = scan_bwd_dual(
        Ia.conv_out, T(a[9]), Ia.A, Ia.bc.B, Ia.bc.C, Ia.D_, Ia.delta_bias_, a[10], T(a[11]), Ia.zeros.narrow(0, 0, Ia.n_scan), Ia.Bv, Ia.Cv,
        Ib.conv_out, T(b[9]), Ib.A, Ib.bc.B, Ib.bc.C, Ib.D_, Ib.delta_bias_, b[10], T(b[11]), Ib.zeros.narrow(0, 0, Ib.n_scan), Ib.Bv, Ib.Cv,
        Ia.z, dout, Ia.dz, delta_softplus, /*keep_fp32=*/true, /*accumulate_dz=*/false, Ia.bc.pad, Ib.bc.pad, impl, segments);
    // d_dt = W_dt^T ddelta of both directions (and the rounding of their dB / dC into dx_dbl) as ONE vms_proj_kred launch: two grids of
    // ~1.5 workgroups per CU each otherwise ((8, 768, 3136): 2 x 11.8 us)
    {
        const Tensor &dda = *g[0][1], &ddb = *g[1][1];
        const int64_t L = Ia.x_dbl.size(2);
        if (Ia.use_kred && Ib.use_kred && L % 4 == 0 && Ia.R == Ib.R && Ia.N == Ib.N && Ia.x_dbl.sizes() == Ib.x_dbl.sizes() &&
            Ia.x_dbl.strides() == Ib.x_dbl.strides() && dda.sizes() == ddb.sizes() && dda.strides() == ddb.strides() &&
            Ia.dt_proj_w.strides() == Ib.dt_proj_w.strides()) {
            Tensor xa = at::empty_like(Ia.x_dbl), xb = at::empty_like(Ib.x_dbl);
            Tensor da = xa.narrow(1, 0, Ia.R), db = xb.narrow(1, 0, Ib.R);
            if (xa.strides() == xb.strides() && proj_kred_eligible(Ia.dt_proj_w.t(), dda, da) && proj_kred_eligible(Ib.dt_proj_w.t(), ddb, db)) {
                const Tensor dbca = Ia.zeros.narrow(0, Ia.A.numel(), 2 * Ia.Bv.numel()).view({2, Ia.b, Ia.N, L});
                const Tensor dbcb = Ib.zeros.narrow(0, Ib.A.numel(), 2 * Ib.Bv.numel()).view({2, Ib.b, Ib.N, L});
                proj_kred(Ia.dt_proj_w.t(), dda, da, Ib.dt_proj_w.t(), ddb, db, 0, dbca, dbcb);
                Ia.dx_dbl_pre = xa;
                Ib.dx_dbl_pre = xb;
            }
        }
    }
    std::vector<OptT> ra = inner_bwd_finish(Ia, g[0]);
    std::vector<OptT> rb = inner_bwd_finish(Ib, g[1]);
    std::vector<OptT> res{ra[0]};
    res.insert(res.end(), ra.begin() + 1, ra.end());
    res.insert(res.end(), rb.begin() + 1, rb.end());
    return res;
}

// reserve: event pairs created now, outside the timed region (the pool grows on demand if a run needs more)
// only: time this entry point alone (two event records per timed launch sit in the stream as markers: ~6 us of idle GPU around
// every launch, 0.1 ms per block step with every launch timed)
void timing_start(int64_t reserve, const std::string& only) {
    std::lock_guard<std::mutex> lk(g_timing_mu);
    g_timing_only = only;
    for (auto& t : g_timed) { g_event_pool.push_back(t.e0); g_event_pool.push_back(t.e1); }
    g_timed.clear();
    g_timed.reserve((size_t)reserve);
    while ((int64_t)g_event_pool.size() < 2 * reserve) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) break;
        g_event_pool.push_back(e);
    }
    g_timing.store(true, std::memory_order_release);
}
// -> [(entry point, milliseconds)] in launch order; synchronises the device.  The events go back to the pool.
std::vector<std::tuple<std::string, double, std::string>> timing_stop_detail() {
    g_timing.store(false, std::memory_order_release);
    (void)hipDeviceSynchronize();
    std::lock_guard<std::mutex> lk(g_timing_mu);
    std::vector<std::tuple<std::string, double, std::string>> out;
    for (auto& t : g_timed) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, t.e0, t.e1);
        out.emplace_back(std::string(t.name), (double)ms, std::string(t.kernel ? t.kernel : ""));
        g_event_pool.push_back(t.e0);
        g_event_pool.push_back(t.e1);
    }
    g_timed.clear();
    return out;
}
std::vector<std::tuple<std::string, double>> timing_stop() {
    std::vector<std::tuple<std::string, double>> out;
    for (auto& t : timing_stop_detail()) out.emplace_back(std::get<0>(t), std::get<1>(t));
    return out;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "compiled PyTorch binding of libvms_hip.so (include/vms_hip.h)";
    namespace py = pybind11;
    m.def("scan_fwd", &scan_fwd, py::arg("u"), py::arg("delta"), py::arg("A"), py::arg("B"), py::arg("C"), py::arg("D"), py::arg("z"),
          py::arg("delta_bias"), py::arg("delta_softplus"), py::arg("reverse"), py::arg("out_z_into"), py::arg("bc_pad"), py::arg("impl"),
          py::arg("segments"), py::arg("reverse_from") = 0, py::arg("x_mode") = -1);
    m.def("scan_bwd", &scan_bwd, py::arg("u"), py::arg("delta"), py::arg("A"), py::arg("B"), py::arg("C"), py::arg("D"), py::arg("z"),
          py::arg("delta_bias"), py::arg("dout"), py::arg("x"), py::arg("out"), py::arg("dz"), py::arg("delta_softplus"),
          py::arg("recompute_out_z"), py::arg("reverse"), py::arg("zeroed"), py::arg("keep_fp32"), py::arg("accumulate_dz"), py::arg("bc_pad"),
          py::arg("impl"), py::arg("segments"), py::arg("Bshape"), py::arg("Cshape"), py::arg("reverse_from") = 0);
    m.def("conv_fwd", &conv_fwd, py::arg("x"), py::arg("weight"), py::arg("bias"), py::arg("silu"), py::arg("reverse"), py::arg("reverse_from") = 0);
    m.def("conv_bwd", &conv_bwd, py::arg("x"), py::arg("weight"), py::arg("bias"), py::arg("dout"), py::arg("dx"), py::arg("silu"),
          py::arg("reverse"), py::arg("zeroed"), py::arg("accumulate_dx"), py::arg("reverse_from") = 0);
    m.def("conv_update", &conv_update);
    m.def("conv_fwd_dual", &conv_fwd_dual);
    m.def("conv_xproj_dual", &conv_xproj_dual);
    m.def("inner_fwd", &inner_fwd, py::arg("xz"), py::arg("conv_w"), py::arg("conv_b"), py::arg("x_proj_w"), py::arg("dt_proj_w"), py::arg("A"),
          py::arg("D"), py::arg("delta_bias"), py::arg("delta_softplus"), py::arg("reverse"), py::arg("out_z_into"), py::arg("impl"),
          py::arg("segments"), py::arg("reverse_from"), py::arg("proj_flags"), py::arg("conv_out_given") = py::none(), py::arg("x_dbl_given") = py::none(), py::arg("seq_valid") = 0);
    m.def("proj_kred", &proj_kred, py::arg("w"), py::arg("inp"), py::arg("out"), py::arg("w2") = py::none(), py::arg("inp2") = py::none(),
          py::arg("out2") = py::none(), py::arg("tile") = 0, py::arg("cast_src") = py::none(), py::arg("cast_src2") = py::none());
    m.def("x_proj_dual", &x_proj_dual);
    m.def("inner_bwd", &inner_bwd);
    m.def("inner_bwd_dual", &inner_bwd_dual);
    m.def("scan_bwd_dual", &scan_bwd_dual);
    m.def("timing_start", &timing_start, pybind11::arg("reserve") = 0, pybind11::arg("only") = "");
    m.def("timing_stop", &timing_stop);
    m.def("timing_stop_detail", &timing_stop_detail);   // + the kernel each launch chose (tools/suite_shapes.py)
    m.def("abi_version", []() { return vms_abi_version(); });
    m.def("last_kernel", []() { return std::string(vms_last_kernel()); });
}
