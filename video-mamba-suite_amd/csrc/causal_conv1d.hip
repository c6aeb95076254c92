// causal_conv1d.hip -- depthwise causal conv1d (+bias, +SiLU) forward / backward / decode step
// for gfx950 (wave64).
//
// Replaces causal_conv1d_fwd_kernel / channellast_fwd (causal-conv1d/csrc/causal_conv1d_fwd.cu:
// 39-130, 193-298), causal_conv1d_bwd_kernel / channellast_bwd (causal_conv1d_bwd.cu:46-240,
// 306-472) and causal_conv1d_update_kernel (causal_conv1d_update.cu:26-66).
//
// Design (DESIGN.md "causal conv1d"): pure HBM streaming.
//   * L-contiguous layout: a lane owns one 16-byte vector (8 x 16-bit or 4 x fp32 elements);
//     the W-1 halo elements come from the neighbouring lane by a DPP wave shift, only the
//     edge lane of each wave touches memory for them -> every byte of x / dout is requested
//     once per wave, no LDS, no barriers (the reference exchanges the halo through shared
//     memory with 3 __syncthreads per chunk, causal_conv1d_fwd.cu:90-97).
//   * channel-last layout: a lane owns 16 bytes of channels and walks a 64-step L segment
//     with a register sliding window; consecutive lanes = consecutive channels (coalesced).
//   * widths 2..4 are run as a 4-tap filter whose leading taps are zero.
//   * dweight / dbias: per-lane partial sums -> DPP wave reduction -> fp32 atomics.
#include "vms_common.cuh"

namespace vms {

constexpr int kConvThreads = 256;
constexpr int kTaps = 4;

__device__ __forceinline__ float load_w(const void* p, int64_t idx, int wdtype) {
    if (wdtype == VMS_F32) return static_cast<const float*>(p)[idx];
    if (wdtype == VMS_F16) return static_cast<float>(static_cast<const f16_t*>(p)[idx]);
    return static_cast<float>(static_cast<const bf16_t*>(p)[idx]);
}

// taps[k], k = 0..3, multiply x[l - 3 + k]; widths < 4 get leading zeros
__device__ __forceinline__ void load_taps(const vms_conv_fwd_params& p, int c, float (&taps)[kTaps], float& bias) {
#pragma unroll
    for (int k = 0; k < kTaps; ++k) {
        const int w = k - (kTaps - p.width);
        taps[k] = w >= 0 ? load_w(p.weight, (int64_t)c * p.weight_c_stride + (int64_t)w * p.weight_width_stride, p.wdtype) : 0.f;
    }
    bias = p.bias ? load_w(p.bias, c, p.wdtype) : 0.f;
}

__device__ __forceinline__ float silu_grad(float pre) {
    const float s = sigmoidf_(pre);
    return s * (1.f + pre * (1.f - s));
}

// ============================ L-contiguous forward ===========================================
template <typename T, bool SILU, bool VEC>
__global__ __launch_bounds__(kConvThreads) void conv_fwd_kernel(const vms_conv_fwd_params p) {
    constexpr int E = 16 / sizeof(T);
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.y, b = blockIdx.z;
    const int L = p.seqlen;
    const bool rev = p.reverse != 0;  // logical position t <-> physical L-1-t (anti-causal filter)
    const int l0 = (blockIdx.x * kConvThreads + threadIdx.x) * E;
    const T* x = static_cast<const T*>(p.x) + (int64_t)b * p.x_batch_stride + (int64_t)c * p.x_c_stride;
    T* out = static_cast<T*>(p.out) + (int64_t)b * p.out_batch_stride + (int64_t)c * p.out_c_stride;
    float taps[kTaps], bias;
    load_taps(p, c, taps, bias);

    float xv[E + 3];  // xv[3 + i] = x[l0 + i]; xv[0..2] = x[l0-3 .. l0-1]
    float cur[E];
    load_dir<T, E, VEC>(x, l0, L, rev, cur);
#pragma unroll
    for (int i = 0; i < E; ++i) xv[3 + i] = cur[i];
    // halo from the previous lane; the first lane of each wave reads it from memory
#pragma unroll
    for (int j = 0; j < 3; ++j) xv[2 - j] = dpp_mov<DPP_WAVE_SHR1, 0xf>(0.f, cur[E - 1 - j]);
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int l = l0 - 1 - j;
            xv[2 - j] = (l >= 0 && l < L) ? static_cast<float>(x[rev ? L - 1 - l : l]) : 0.f;
        }
    }
    float o[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
        float acc = bias;
#pragma unroll
        for (int k = 0; k < kTaps; ++k) acc = fmaf(taps[k], xv[i + k], acc);
        o[i] = SILU ? acc * sigmoidf_(acc) : acc;
    }
    store_dir<T, E, VEC>(out, l0, L, rev, o);
}

// ============================ L-contiguous backward ==========================================
template <typename T, bool SILU, bool VEC>
__global__ __launch_bounds__(kConvThreads) void conv_bwd_kernel(const vms_conv_bwd_params q) {
    const vms_conv_fwd_params& p = q.f;
    constexpr int E = 16 / sizeof(T);
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.y, b = blockIdx.z;
    const int L = p.seqlen;
    const bool rev = p.reverse != 0;  // logical position t <-> physical L-1-t (anti-causal filter)
    const int l0 = (blockIdx.x * kConvThreads + threadIdx.x) * E;
    const T* x = static_cast<const T*>(p.x) + (int64_t)b * p.x_batch_stride + (int64_t)c * p.x_c_stride;
    const T* dout = static_cast<const T*>(q.dout) + (int64_t)b * q.dout_batch_stride + (int64_t)c * q.dout_c_stride;
    T* dx = static_cast<T*>(q.dx) + (int64_t)b * q.dx_batch_stride + (int64_t)c * q.dx_c_stride;
    float taps[kTaps], bias;
    load_taps(p, c, taps, bias);

    float xv[E + 3], cur[E], gc[E], g[E + 3];  // g[i] = dout'[l0 + i], i up to E+2 (right halo)
    load_dir<T, E, VEC>(x, l0, L, rev, cur);
    load_dir<T, E, VEC>(dout, l0, L, rev, gc);
#pragma unroll
    for (int i = 0; i < E; ++i) { xv[3 + i] = cur[i]; g[i] = gc[i]; }
#pragma unroll
    for (int j = 0; j < 3; ++j) xv[2 - j] = dpp_mov<DPP_WAVE_SHR1, 0xf>(0.f, cur[E - 1 - j]);
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int l = l0 - 1 - j;
            xv[2 - j] = (l >= 0 && l < L) ? static_cast<float>(x[rev ? L - 1 - l : l]) : 0.f;
        }
    }
    if (SILU) {
#pragma unroll
        for (int i = 0; i < E; ++i) {
            float pre = bias;
#pragma unroll
            for (int k = 0; k < kTaps; ++k) pre = fmaf(taps[k], xv[i + k], pre);
            g[i] *= silu_grad(pre);
        }
    }
    // right halo dout'[l0+E .. l0+E+2] from the next lane; the last lane recomputes it
#pragma unroll
    for (int j = 0; j < 3; ++j) g[E + j] = dpp_mov<DPP_WAVE_SHL1, 0xf>(0.f, g[j]);
    if (lane == 63) {
        float xn[6];  // x[l0+E-3 .. l0+E+2]
#pragma unroll
        for (int j = 0; j < 3; ++j) xn[j] = cur[E - 3 + j];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int l = l0 + E + j;
            xn[3 + j] = l < L ? static_cast<float>(x[rev ? L - 1 - l : l]) : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int l = l0 + E + j;
            float go = l < L ? static_cast<float>(dout[rev ? L - 1 - l : l]) : 0.f;
            if (SILU) {
                float pre = bias;
#pragma unroll
                for (int k = 0; k < kTaps; ++k) pre = fmaf(taps[k], xn[j + k], pre);
                go *= silu_grad(pre);
            }
            g[E + j] = go;
        }
    }
    // dx[l] = sum_k taps[k] * dout'[l + 3 - k]
    float o[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < kTaps; ++k) acc = fmaf(taps[k], g[i + 3 - k], acc);
        o[i] = acc;
    }
    store_dir<T, E, VEC>(dx, l0, L, rev, o);
    // dweight[k] += x[l - 3 + k] * dout'[l] ; dbias += dout'[l]   (own elements only)
    float dw[kTaps] = {0.f, 0.f, 0.f, 0.f}, db = 0.f;
#pragma unroll
    for (int i = 0; i < E; ++i) {
        db += g[i];
#pragma unroll
        for (int k = 0; k < kTaps; ++k) dw[k] = fmaf(xv[i + k], g[i], dw[k]);
    }
#pragma unroll
    for (int k = 0; k < kTaps; ++k) {
        const int w = k - (kTaps - p.width);
        const float t = wave_sum(dw[k]);
        if (lane == 0 && w >= 0) atomicAdd(q.dweight + (int64_t)c * q.dweight_c_stride + (int64_t)w * q.dweight_width_stride, t);
    }
    if (q.dbias) {
        const float t = wave_sum(db);
        if (lane == 0) atomicAdd(q.dbias + c, t);
    }
}

// ============================ channel-last kernels ===========================================
// memory order (batch, L, dim): x_c_stride == 1.  lane = 16-byte channel vector, wave = L segment
constexpr int kSegL = 64;

template <typename T, bool SILU>
__global__ __launch_bounds__(kConvThreads) void conv_cl_fwd_kernel(const vms_conv_fwd_params p) {
    constexpr int E = 16 / sizeof(T);
    using V = vec_t<T, E>;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = (blockIdx.x * 64 + lane) * E;
    const int s0 = (blockIdx.y * 4 + wave) * kSegL;
    const int b = blockIdx.z, L = p.seqlen;
    if (c0 >= p.dim || s0 >= L) return;
    const T* x = static_cast<const T*>(p.x) + (int64_t)b * p.x_batch_stride + c0;
    T* out = static_cast<T*>(p.out) + (int64_t)b * p.out_batch_stride + c0;
    float taps[E][kTaps], bias[E];
#pragma unroll
    for (int e = 0; e < E; ++e) load_taps(p, c0 + e, taps[e], bias[e]);
    float win[kTaps][E];  // win[k] = x[l - 3 + k]
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int l = s0 - 3 + k;
        V v = {};
        if (l >= 0) v = *reinterpret_cast<const V*>(x + (int64_t)l * p.x_l_stride);
#pragma unroll
        for (int e = 0; e < E; ++e) win[k + 1][e] = static_cast<float>(v[e]);
    }
    const int s1 = min(s0 + kSegL, L);
    for (int l = s0; l < s1; ++l) {
        V v = *reinterpret_cast<const V*>(x + (int64_t)l * p.x_l_stride);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            win[0][e] = win[1][e]; win[1][e] = win[2][e]; win[2][e] = win[3][e];
            win[3][e] = static_cast<float>(v[e]);
        }
        V o;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            float acc = bias[e];
#pragma unroll
            for (int k = 0; k < kTaps; ++k) acc = fmaf(taps[e][k], win[k][e], acc);
            o[e] = static_cast<T>(SILU ? acc * sigmoidf_(acc) : acc);
        }
        *reinterpret_cast<V*>(out + (int64_t)l * p.out_l_stride) = o;
    }
}

template <typename T, bool SILU>
__global__ __launch_bounds__(kConvThreads) void conv_cl_bwd_kernel(const vms_conv_bwd_params q) {
    const vms_conv_fwd_params& p = q.f;
    constexpr int E = 16 / sizeof(T);
    using V = vec_t<T, E>;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = (blockIdx.x * 64 + lane) * E;
    const int s0 = (blockIdx.y * 4 + wave) * kSegL;
    const int b = blockIdx.z, L = p.seqlen;
    if (c0 >= p.dim || s0 >= L) return;
    const T* x = static_cast<const T*>(p.x) + (int64_t)b * p.x_batch_stride + c0;
    const T* dout = static_cast<const T*>(q.dout) + (int64_t)b * q.dout_batch_stride + c0;
    T* dx = static_cast<T*>(q.dx) + (int64_t)b * q.dx_batch_stride + c0;
    float taps[E][kTaps], bias[E];
#pragma unroll
    for (int e = 0; e < E; ++e) load_taps(p, c0 + e, taps[e], bias[e]);
    float xw[kTaps][E], gw[kTaps][E];  // xw[k] = x[t-3+k]; gw[k] = dout'[t-3+k]
    float dw[E][kTaps], db[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        db[e] = 0.f;
#pragma unroll
        for (int k = 0; k < kTaps; ++k) { dw[e][k] = 0.f; gw[k][e] = 0.f; xw[k][e] = 0.f; }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int l = s0 - 3 + k;
        V v = {};
        if (l >= 0) v = *reinterpret_cast<const V*>(x + (int64_t)l * p.x_l_stride);
#pragma unroll
        for (int e = 0; e < E; ++e) xw[k + 1][e] = static_cast<float>(v[e]);
    }
    const int s1 = min(s0 + kSegL, L);
    // forward sweep over t; dx[t-3] becomes final once dout'[t] is known
    for (int t = s0; t < s1 + 3; ++t) {
        V xv = {}, gv = {};
        if (t < L) {
            xv = *reinterpret_cast<const V*>(x + (int64_t)t * p.x_l_stride);
            gv = *reinterpret_cast<const V*>(dout + (int64_t)t * q.dout_l_stride);
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            xw[0][e] = xw[1][e]; xw[1][e] = xw[2][e]; xw[2][e] = xw[3][e];
            xw[3][e] = static_cast<float>(xv[e]);
            gw[0][e] = gw[1][e]; gw[1][e] = gw[2][e]; gw[2][e] = gw[3][e];
            float go = static_cast<float>(gv[e]);
            if (SILU) {
                float pre = bias[e];
#pragma unroll
                for (int k = 0; k < kTaps; ++k) pre = fmaf(taps[e][k], xw[k][e], pre);
                go *= silu_grad(pre);
            }
            gw[3][e] = go;
            if (t < s1) {  // own element: weight / bias gradients
                db[e] += go;
#pragma unroll
                for (int k = 0; k < kTaps; ++k) dw[e][k] = fmaf(xw[k][e], go, dw[e][k]);
            }
        }
        const int l = t - 3;
        if (l >= s0 && l < s1) {
            V o;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                float acc = 0.f;  // dx[l] = sum_k taps[k] * dout'[l + 3 - k] = sum_k taps[k] * gw[3-k]
#pragma unroll
                for (int k = 0; k < kTaps; ++k) acc = fmaf(taps[e][k], gw[3 - k][e], acc);
                o[e] = static_cast<T>(acc);
            }
            *reinterpret_cast<V*>(dx + (int64_t)l * q.dx_l_stride) = o;
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
#pragma unroll
        for (int k = 0; k < kTaps; ++k) {
            const int w = k - (kTaps - p.width);
            if (w >= 0) atomicAdd(q.dweight + (int64_t)(c0 + e) * q.dweight_c_stride + (int64_t)w * q.dweight_width_stride, dw[e][k]);
        }
        if (q.dbias) atomicAdd(q.dbias + c0 + e, db[e]);
    }
}

// ============================ decode step ====================================================
template <typename T>
__global__ __launch_bounds__(64) void conv_update_kernel(const vms_conv_fwd_params p) {
    const int c = blockIdx.y * 64 + threadIdx.x, b = blockIdx.x;
    if (c >= p.dim) return;
    T* cs = static_cast<T*>(p.conv_state) + (int64_t)b * p.conv_state_batch_stride + (int64_t)c * p.conv_state_c_stride;
    const T xin = static_cast<const T*>(p.x)[(int64_t)b * p.x_batch_stride + (int64_t)c * p.x_c_stride];
    float acc = p.bias ? load_w(p.bias, c, p.wdtype) : 0.f;
    for (int w = 0; w < p.width; ++w) {
        const T v = w + 1 < p.width ? cs[(int64_t)(w + 1) * p.conv_state_l_stride] : xin;
        cs[(int64_t)w * p.conv_state_l_stride] = v;  // shift left by one, append x (bit-exact)
        acc = fmaf(load_w(p.weight, (int64_t)c * p.weight_c_stride + (int64_t)w * p.weight_width_stride, p.wdtype),
                   static_cast<float>(v), acc);
    }
    if (p.silu_activation) acc = acc * sigmoidf_(acc);
    static_cast<T*>(p.out)[(int64_t)b * p.out_batch_stride + (int64_t)c * p.out_c_stride] = static_cast<T>(acc);
}

// ============================ host side ======================================================
static int validate_conv(const vms_conv_fwd_params& p) {
    VMS_CHECK(p.dtype == VMS_F32 || p.dtype == VMS_F16 || p.dtype == VMS_BF16, "input dtype must be fp32/fp16/bf16");
    VMS_CHECK(p.wdtype == VMS_F32 || p.wdtype == VMS_F16 || p.wdtype == VMS_BF16, "weight dtype must be fp32/fp16/bf16");
    VMS_CHECK(p.batch > 0 && p.dim > 0 && p.seqlen > 0, "empty problem");
    VMS_CHECK(p.width >= 2 && p.width <= 4, "causal_conv1d only supports width between 2 and 4");
    VMS_CHECK(p.x && p.weight, "x and weight are required");
    VMS_CHECK(!p.reverse || (p.x_l_stride == 1 && !p.conv_state), "reverse (anti-causal) conv1d needs the seqlen-contiguous layout");
    return VMS_OK;
}

template <typename T>
static int conv_fwd_dispatch(const vms_conv_fwd_params& p, hipStream_t s) {
    constexpr int E = 16 / sizeof(T);
    const int es = sizeof(T);
    const bool channel_last = p.x_c_stride == 1 && p.x_l_stride > 1;
    if (!channel_last) {
        VMS_CHECK(p.x_l_stride == 1 && p.out_l_stride == 1, "x and out need a unit seqlen stride");
        const bool vec = aligned16(p.x) && aligned16(p.out) && mult16(p.x_batch_stride, es) && mult16(p.x_c_stride, es) &&
                         mult16(p.out_batch_stride, es) && mult16(p.out_c_stride, es);
        dim3 grid((p.seqlen + kConvThreads * E - 1) / (kConvThreads * E), p.dim, p.batch), block(kConvThreads);
#define VMS_L(S_, V_) hipLaunchKernelGGL((conv_fwd_kernel<T, S_, V_>), grid, block, 0, s, p)
        if (p.silu_activation) { if (vec) VMS_L(true, true); else VMS_L(true, false); }
        else { if (vec) VMS_L(false, true); else VMS_L(false, false); }
#undef VMS_L
    } else {
        VMS_CHECK(p.dim % 8 == 0, "causal_conv1d only supports channel dimension divisible by 8 for now");
        VMS_CHECK(p.out_c_stride == 1, "channel-last x needs a channel-last out");
        VMS_CHECK(aligned16(p.x) && aligned16(p.out) && mult16(p.x_batch_stride, es) && mult16(p.x_l_stride, es) &&
                      mult16(p.out_batch_stride, es) && mult16(p.out_l_stride, es),
                  "channel-last tensors must be 16-byte aligned in batch / seqlen strides");
        dim3 grid((p.dim / E + 63) / 64, (p.seqlen + 4 * kSegL - 1) / (4 * kSegL), p.batch), block(kConvThreads);
        if (p.silu_activation) hipLaunchKernelGGL((conv_cl_fwd_kernel<T, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((conv_cl_fwd_kernel<T, false>), grid, block, 0, s, p);
    }
    VMS_LAUNCH_CHECK();
    return VMS_OK;
}

template <typename T>
static int conv_bwd_dispatch(const vms_conv_bwd_params& q, hipStream_t s) {
    const vms_conv_fwd_params& p = q.f;
    constexpr int E = 16 / sizeof(T);
    const int es = sizeof(T);
    const bool channel_last = p.x_c_stride == 1 && p.x_l_stride > 1;
    if (!channel_last) {
        VMS_CHECK(p.x_l_stride == 1 && q.dout_l_stride == 1 && q.dx_l_stride == 1, "x, dout, dx need a unit seqlen stride");
        const bool vec = aligned16(p.x) && aligned16(q.dout) && aligned16(q.dx) && mult16(p.x_batch_stride, es) &&
                         mult16(p.x_c_stride, es) && mult16(q.dout_batch_stride, es) && mult16(q.dout_c_stride, es) &&
                         mult16(q.dx_batch_stride, es) && mult16(q.dx_c_stride, es);
        dim3 grid((p.seqlen + kConvThreads * E - 1) / (kConvThreads * E), p.dim, p.batch), block(kConvThreads);
#define VMS_L(S_, V_) hipLaunchKernelGGL((conv_bwd_kernel<T, S_, V_>), grid, block, 0, s, q)
        if (p.silu_activation) { if (vec) VMS_L(true, true); else VMS_L(true, false); }
        else { if (vec) VMS_L(false, true); else VMS_L(false, false); }
#undef VMS_L
    } else {
        VMS_CHECK(p.dim % 8 == 0, "causal_conv1d only supports channel dimension divisible by 8 for now");
        VMS_CHECK(q.dout_c_stride == 1 && q.dx_c_stride == 1, "channel-last x needs channel-last dout and dx");
        VMS_CHECK(aligned16(p.x) && aligned16(q.dout) && aligned16(q.dx) && mult16(p.x_batch_stride, es) &&
                      mult16(p.x_l_stride, es) && mult16(q.dout_batch_stride, es) && mult16(q.dout_l_stride, es) &&
                      mult16(q.dx_batch_stride, es) && mult16(q.dx_l_stride, es),
                  "channel-last tensors must be 16-byte aligned in batch / seqlen strides");
        dim3 grid((p.dim / E + 63) / 64, (p.seqlen + 4 * kSegL - 1) / (4 * kSegL), p.batch), block(kConvThreads);
        if (p.silu_activation) hipLaunchKernelGGL((conv_cl_bwd_kernel<T, true>), grid, block, 0, s, q);
        else hipLaunchKernelGGL((conv_cl_bwd_kernel<T, false>), grid, block, 0, s, q);
    }
    VMS_LAUNCH_CHECK();
    return VMS_OK;
}

}  // namespace vms

using namespace vms;

extern "C" int vms_causal_conv1d_fwd(const vms_conv_fwd_params* pp, void* stream) {
    VMS_CHECK(pp != nullptr, "null params");
    if (int rc = validate_conv(*pp)) return rc;
    VMS_CHECK(pp->out != nullptr, "out must be provided by the caller");
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (pp->dtype) {
        case VMS_F32: return conv_fwd_dispatch<float>(*pp, s);
        case VMS_F16: return conv_fwd_dispatch<f16_t>(*pp, s);
        default: return conv_fwd_dispatch<bf16_t>(*pp, s);
    }
}

extern "C" int vms_causal_conv1d_bwd(const vms_conv_bwd_params* qq, void* stream) {
    VMS_CHECK(qq != nullptr, "null params");
    if (int rc = validate_conv(qq->f)) return rc;
    VMS_CHECK(qq->dout && qq->dx && qq->dweight, "dout, dx and dweight are required");
    VMS_CHECK((qq->f.bias == nullptr) == (qq->dbias == nullptr), "dbias must be given iff bias is given");
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (qq->f.dtype) {
        case VMS_F32: return conv_bwd_dispatch<float>(*qq, s);
        case VMS_F16: return conv_bwd_dispatch<f16_t>(*qq, s);
        default: return conv_bwd_dispatch<bf16_t>(*qq, s);
    }
}

extern "C" int vms_causal_conv1d_update(const vms_conv_fwd_params* pp, void* stream) {
    VMS_CHECK(pp != nullptr, "null params");
    vms_conv_fwd_params p = *pp;
    p.seqlen = 1;
    if (int rc = validate_conv(p)) return rc;
    VMS_CHECK(p.out && p.conv_state, "out and conv_state are required");
    hipStream_t s = static_cast<hipStream_t>(stream);
    dim3 grid(p.batch, (p.dim + 63) / 64), block(64);
    switch (p.dtype) {
        case VMS_F32: hipLaunchKernelGGL((conv_update_kernel<float>), grid, block, 0, s, p); break;
        case VMS_F16: hipLaunchKernelGGL((conv_update_kernel<f16_t>), grid, block, 0, s, p); break;
        default: hipLaunchKernelGGL((conv_update_kernel<bf16_t>), grid, block, 0, s, p); break;
    }
    VMS_LAUNCH_CHECK();
    return VMS_OK;
}
