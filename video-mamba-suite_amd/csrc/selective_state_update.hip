// selective_state_update.hip -- single-token SSM step for gfx950 (wave64).
//
// Replaces the Triton kernel _selective_scan_update_kernel behind selective_state_update
// (mamba/mamba_ssm/ops/triton/selective_state_update.py:16-154; semantics of selective_state_update_ref
// :157-192): dt = softplus?(dt + dt_bias); state = state * exp(dt * A) + dt * B * x (in place);
// out = sum_n state * C + D * x; out *= silu(z).  Only Mamba.step (autoregressive decode) calls it; no video
// task does -- it completes the extension surface.
// Layout: 16 lanes per (batch, channel) row, lane j walks the states j, j + 16, ...; a wave = 4 rows; the
// contraction with C is a 4-step DPP rotation sum inside the 16-lane row.  Memory-bound on `state`
// (read + written once), everything else is per-row scalars.
#include "vms_common.h"

namespace vms {

__device__ __forceinline__ float ld_any(const void* p, int64_t i, int dt) {
    if (dt == VMS_F32) return static_cast<const float*>(p)[i];
    if (dt == VMS_F16) return static_cast<float>(static_cast<const f16_t*>(p)[i]);
    return static_cast<float>(static_cast<const bf16_t*>(p)[i]);
}
__device__ __forceinline__ void st_any(void* p, int64_t i, int dt, float v) {
    if (dt == VMS_F32) static_cast<float*>(p)[i] = v;
    else if (dt == VMS_F16) static_cast<f16_t*>(p)[i] = static_cast<f16_t>(v);
    else static_cast<bf16_t*>(p)[i] = static_cast<bf16_t>(v);
}

__global__ __launch_bounds__(256) void state_update_kernel(const vms_state_update_params p) {
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;
    const int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);  // b * dim + d
    const bool ok = row < (int64_t)p.batch * p.dim;
    const int b = ok ? (int)(row / p.dim) : 0, d = ok ? (int)(row % p.dim) : 0;
    float t = ld_any(p.dt, (int64_t)b * p.dt_batch_stride + (int64_t)d * p.dt_d_stride, p.dt_dtype);
    if (p.dt_bias) t += ld_any(p.dt_bias, d, p.w_dtype);
    if (p.dt_softplus) t = softplusf_(t);
    const float xv = ld_any(p.x, (int64_t)b * p.x_batch_stride + (int64_t)d * p.x_d_stride, p.x_dtype);
    const float tx = t * xv;
    float acc = 0.f;
    for (int n = j; n < p.dstate; n += 16) {
        const int64_t si = (int64_t)b * p.state_batch_stride + (int64_t)d * p.state_d_stride + (int64_t)n * p.state_n_stride;
        const float a = fast_exp(t * ld_any(p.A, (int64_t)d * p.A_d_stride + (int64_t)n * p.A_n_stride, p.w_dtype));
        const float Bn = ld_any(p.B, (int64_t)b * p.B_batch_stride + (int64_t)n * p.B_n_stride, p.bc_dtype);
        const float Cn = ld_any(p.C, (int64_t)b * p.C_batch_stride + (int64_t)n * p.C_n_stride, p.bc_dtype);
        float s = fmaf(ld_any(p.state, si, p.state_dtype), a, tx * Bn);
        if (ok) st_any(p.state, si, p.state_dtype, s);
        if (p.state_dtype != VMS_F32)  // the contraction sees the stored (rounded) state, like the reference
            s = p.state_dtype == VMS_F16 ? static_cast<float>(static_cast<f16_t>(s)) : static_cast<float>(static_cast<bf16_t>(s));
        acc = fmaf(s, Cn, acc);
    }
    acc += dpp_mov<0x121, 0xf>(0.f, acc);  // row_ror:1, 2, 4, 8: sum over the row's 16 lanes
    acc += dpp_mov<0x122, 0xf>(0.f, acc);
    acc += dpp_mov<0x124, 0xf>(0.f, acc);
    acc += dpp_mov<0x128, 0xf>(0.f, acc);
    if (p.D) acc = fmaf(xv, ld_any(p.D, d, p.w_dtype), acc);
    if (p.z) {
        const float zv = ld_any(p.z, (int64_t)b * p.z_batch_stride + (int64_t)d * p.z_d_stride, p.z_dtype);
        acc *= zv * sigmoidf_(zv);
    }
    if (ok && j == 0) st_any(p.out, (int64_t)b * p.out_batch_stride + (int64_t)d * p.out_d_stride, p.x_dtype, acc);
}

}  // namespace vms

using namespace vms;

extern "C" int vms_selective_state_update(const vms_state_update_params* pp, void* stream) {
    VMS_CHECK(pp != nullptr, "null params");
    const vms_state_update_params& p = *pp;
    auto dt_ok = [](int d) { return d == VMS_F32 || d == VMS_F16 || d == VMS_BF16; };
    VMS_CHECK(dt_ok(p.state_dtype) && dt_ok(p.x_dtype) && dt_ok(p.bc_dtype) && dt_ok(p.w_dtype) && dt_ok(p.dt_dtype) && (!p.z || dt_ok(p.z_dtype)),
              "dtypes must be fp32/fp16/bf16");
    VMS_CHECK(p.batch > 0 && p.dim > 0 && p.dstate > 0, "empty problem");
    VMS_CHECK(p.state && p.x && p.dt && p.A && p.B && p.C && p.out, "state, x, dt, A, B, C, out are required");
    const int64_t rows = (int64_t)p.batch * p.dim;
    hipLaunchKernelGGL(state_update_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    VMS_LAUNCH_CHECK();
    return VMS_OK;
}
extern "C" int vms_sizeof_state_update_params(void) { return (int)sizeof(vms_state_update_params); }
