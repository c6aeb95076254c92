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
#include "vms_common.h"

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

// ============================ L-contiguous kernels ===========================================
// A wave owns NV consecutive strips of 64 * E elements of one (batch, channel) row; a lane owns one
// 16-byte vector in each.  All NV (forward) / 2 NV (backward) loads of a lane are in flight together --
// with a single vector per lane a CU holds 32 KB of requests, half of what 8 TB/s x ~2 us needs -- and the
// strip boundaries inside the wave are crossed with v_readlane instead of memory.
// DIR selects the addressing: 0 = generic (run-time direction, element-wise tails, any alignment);
// 1 / 2 = left-to-right / right-to-left over 16-byte aligned rows with seqlen % E == 0, through a buffer
// resource per row: out-of-range lanes read zeros and their stores are dropped by the bounds check, so
// the hot path has no predication at all.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kBufFlags = 0x00020000;  // gfx9 raw buffer, 32-bit data format

__device__ __forceinline__ float lane_value(float v, int src_lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src_lane));
}

// RAGC: rows that are not 16-byte aligned and / or seqlen % E != 0.  gfx950 serves 16-byte buffer accesses at any
// 2-byte alignment; the row's single partly valid vector moves element by element (one lane, once per row).
template <typename T, int DIR, bool RAGC>
struct RowIO {
    static constexpr int E = 16 / sizeof(T);
    __amdgpu_buffer_rsrc_t rsrc;
    int L;
    __device__ __forceinline__ RowIO(const T* row, int seqlen) : L(seqlen) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(row), 0, seqlen * (int)sizeof(T), kBufFlags);
    }
    // byte offset of the vector holding logical elements [l0, l0 + E); negative = out of range
    __device__ __forceinline__ int voff(int l0) const { return (DIR == 2 ? L - l0 - E : l0) * (int)sizeof(T); }
    __device__ __forceinline__ void load(int l0, float (&out)[E]) const {
        if (!RAGC || l0 + E <= L || l0 >= L) {
            const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff(l0), 0, 0);
            const vec_t<T, E> t = __builtin_bit_cast(vec_t<T, E>, raw);
#pragma unroll
            for (int e = 0; e < E; ++e) out[DIR == 2 ? E - 1 - e : e] = static_cast<float>(t[e]);
        } else {
#pragma unroll
            for (int i = 0; i < E; ++i) out[i] = elem(l0 + i, true);
        }
    }
    __device__ __forceinline__ void store(int l0, const float (&in)[E]) const {
        if (!RAGC || l0 + E <= L || l0 >= L) {
            vec_t<T, E> t;
#pragma unroll
            for (int e = 0; e < E; ++e) t[e] = static_cast<T>(in[DIR == 2 ? E - 1 - e : e]);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, t), rsrc, voff(l0), 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < E; ++i) {   // stores past the row are dropped by the bounds check
                const int phys = DIR == 2 ? L - 1 - (l0 + i) : l0 + i;
                const T v = static_cast<T>(in[i]);
                if constexpr (sizeof(T) == 2)
                    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, v), rsrc, phys * 2, 0, 0);
                else
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), rsrc, phys * 4, 0, 0);
            }
        }
    }
    // one logical element, 0 outside [0, L); lanes with `active` false do not touch memory
    __device__ __forceinline__ float elem(int l, bool active) const {
        const int phys = DIR == 2 ? L - 1 - l : l;
        const int off = active ? phys * (int)sizeof(T) : -1;
        if constexpr (sizeof(T) == 2) {
            const unsigned short h = __builtin_amdgcn_raw_buffer_load_b16(rsrc, off, 0, 0);
            return static_cast<float>(__builtin_bit_cast(T, h));
        } else {
            return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0));
        }
    }
};

template <typename T, bool VEC>
struct RowIOGeneric {
    static constexpr int E = 16 / sizeof(T);
    T* row;
    int L;
    bool rev;
    __device__ __forceinline__ RowIOGeneric(const T* r, int seqlen, bool reverse) : row(const_cast<T*>(r)), L(seqlen), rev(reverse) {}
    __device__ __forceinline__ void load(int l0, float (&out)[E]) const { load_dir<T, E, VEC>(row, l0, L, rev, out); }
    __device__ __forceinline__ void store(int l0, const float (&in)[E]) const { store_dir<T, E, VEC>(row, l0, L, rev, in); }
    __device__ __forceinline__ float elem(int l, bool active) const {
        return (active && l >= 0 && l < L) ? static_cast<float>(row[rev ? L - 1 - l : l]) : 0.f;
    }
};

// DIR 1 / 2: VEC = false selects the ragged / unaligned flavour of the buffer-addressed rows
template <typename T, bool VEC, int DIR>
struct RowIOSel { using type = RowIO<T, DIR, !VEC>; };
template <typename T, bool VEC>
struct RowIOSel<T, VEC, 0> { using type = RowIOGeneric<T, VEC>; };

template <typename T, bool VEC, int DIR>
__device__ __forceinline__ typename RowIOSel<T, VEC, DIR>::type make_row(const T* row, int L, bool rev) {
    if constexpr (DIR == 0) return RowIOGeneric<T, VEC>(row, L, rev);
    else return RowIO<T, DIR, !VEC>(row, L);
}

template <typename T, bool SILU, bool VEC, int NV, int DIR>
__global__ __launch_bounds__(kConvThreads) void conv_fwd_kernel(const vms_conv_fwd_params p) {
    constexpr int E = 16 / sizeof(T);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int c = blockIdx.y, b = blockIdx.z;
    const int L = p.seqlen;
    const bool rev = p.reverse != 0;  // logical position t <-> physical L-1-t (anti-causal filter)
    const int base = (blockIdx.x * (kConvThreads / 64) + wave) * (64 * E * NV) + lane * E;
    const auto x = make_row<T, VEC, DIR>(static_cast<const T*>(p.x) + (int64_t)b * p.x_batch_stride + (int64_t)c * p.x_c_stride, L, rev);
    const auto out = make_row<T, VEC, DIR>(static_cast<T*>(p.out) + (int64_t)b * p.out_batch_stride + (int64_t)c * p.out_c_stride, L, rev);
    float taps[kTaps], bias;
    load_taps(p, c, taps, bias);

    float xv[NV][E + 3];  // xv[s][3 + i] = x[l0_s + i]; xv[s][0..2] = x[l0_s-3 .. l0_s-1]
#pragma unroll
    for (int s = 0; s < NV; ++s) {
        float cur[E];
        x.load(base + s * 64 * E, cur);
#pragma unroll
        for (int i = 0; i < E; ++i) xv[s][3 + i] = cur[i];
    }
    // halo from the previous lane; lane 0 takes it from lane 63 of the previous strip, or from memory
    float edge[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) edge[j] = x.elem(base - 1 - j, lane == 0);
#pragma unroll
    for (int s = 0; s < NV; ++s) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float own = xv[s][3 + E - 1 - j];
            const float prev = dpp_mov<DPP_WAVE_SHR1, 0xf>(0.f, own);
            const float left = s == 0 ? edge[j] : lane_value(xv[s - 1][3 + E - 1 - j], 63);
            xv[s][2 - j] = lane == 0 ? left : prev;
        }
    }
#pragma unroll
    for (int s = 0; s < NV; ++s) {
        float o[E];
#pragma unroll
        for (int i = 0; i < E; ++i) {
            float acc = bias;
#pragma unroll
            for (int k = 0; k < kTaps; ++k) acc = fmaf(taps[k], xv[s][i + k], acc);
            // the fp32 result is final BEFORE it is narrowed: left to itself hipcc folds the last fma into an fp16 conversion
            // (v_fma_mixlo_f16: one rounding instead of two) in one kernel and not in another -- conv_fwd_dual_kernel must give
            // this kernel's bits for every dtype (vms_hip.h)
            float r = SILU ? acc * sigmoidf_(acc) : acc;
            asm volatile("" : "+v"(r));
            o[i] = r;
        }
        out.store(base + s * 64 * E, o);
    }
}

// Both directions of a bidirectional block in one pass over x (vms_hip.h vms_causal_conv1d_fwd_dual): out = the causal filter
// (weight, bias) and out_b = the ANTI-causal filter (weight_b, bias_b) of the same rows -- what the reference computes as
// conv(x) and flip(conv_b(flip(x))) (mamba_simple.py:244-258).  x is read once (402 instead of 537 MB at (8, 1024, 8192)); both
// outputs are in physical order.  The anti-causal taps run in the order of the right-to-left kernel (DIR = 2 above), so that
// out_b is bit-identical to a `reverse` call: out_b[p] = bias_b + sum_k taps_b[k] x[p + 3 - k].
template <typename T, bool SILU, bool VEC, int NV>
__global__ __launch_bounds__(kConvThreads) void conv_fwd_dual_kernel(const vms_conv_fwd_dual_params q) {
    const vms_conv_fwd_params& p = q.f;
    constexpr int E = 16 / sizeof(T);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int c = blockIdx.y, b = blockIdx.z;
    const int L = p.seqlen;
    const int base = (blockIdx.x * (kConvThreads / 64) + wave) * (64 * E * NV) + lane * E;
    const RowIO<T, 1, !VEC> x(static_cast<const T*>(p.x) + (int64_t)b * p.x_batch_stride + (int64_t)c * p.x_c_stride, L);
    const RowIO<T, 1, !VEC> out(static_cast<T*>(p.out) + (int64_t)b * p.out_batch_stride + (int64_t)c * p.out_c_stride, L);
    const RowIO<T, 1, !VEC> out_b(static_cast<T*>(q.out_b) + (int64_t)b * q.out_b_batch_stride + (int64_t)c * q.out_b_c_stride, L);
    float taps[kTaps], bias, taps_b[kTaps], bias_b;
    load_taps(p, c, taps, bias);
#pragma unroll
    for (int k = 0; k < kTaps; ++k) {
        const int w = k - (kTaps - p.width);
        taps_b[k] = w >= 0 ? load_w(q.weight_b, (int64_t)c * q.weight_b_c_stride + (int64_t)w * q.weight_b_width_stride, p.wdtype) : 0.f;
    }
    bias_b = q.bias_b ? load_w(q.bias_b, c, p.wdtype) : 0.f;

    float xv[NV][E + 6];  // xv[s][3 + i] = x[l0_s + i]; [0..2] = the 3 elements before, [E + 3 .. E + 5] = the 3 after
#pragma unroll
    for (int s = 0; s < NV; ++s) {
        float cur[E];
        x.load(base + s * 64 * E, cur);
#pragma unroll
        for (int i = 0; i < E; ++i) xv[s][3 + i] = cur[i];
    }
    const int lend = base + (NV - 1) * 64 * E + E;   // what follows this lane's last vector (used by lane 63: the wave's end)
    float edge[3], edge_r[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        edge[j] = x.elem(base - 1 - j, lane == 0);
        edge_r[j] = x.elem(lend + j, lane == 63);
    }
#pragma unroll
    for (int s = 0; s < NV; ++s) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float own = xv[s][3 + E - 1 - j];
            const float prev = dpp_mov<DPP_WAVE_SHR1, 0xf>(0.f, own);
            const float left = s == 0 ? edge[j] : lane_value(xv[s - 1][3 + E - 1 - j], 63);
            xv[s][2 - j] = lane == 0 ? left : prev;
            const float own_r = xv[s][3 + j];
            const float next = dpp_mov<DPP_WAVE_SHL1, 0xf>(0.f, own_r);
            const float right = s == NV - 1 ? edge_r[j] : lane_value(xv[s + 1 < NV ? s + 1 : s][3 + j], 0);
            xv[s][E + 3 + j] = lane == 63 ? right : next;
        }
    }
#pragma unroll
    for (int s = 0; s < NV; ++s) {
        float o[E], ob[E];
#pragma unroll
        for (int i = 0; i < E; ++i) {
            float acc = bias, acc_b = bias_b;
#pragma unroll
            for (int k = 0; k < kTaps; ++k) {
                acc = fmaf(taps[k], xv[s][i + k], acc);
                acc_b = fmaf(taps_b[k], xv[s][i + 6 - k], acc_b);
            }
            float r = SILU ? acc * sigmoidf_(acc) : acc, rb = SILU ? acc_b * sigmoidf_(acc_b) : acc_b;
            asm volatile("" : "+v"(r), "+v"(rb));   // as conv_fwd_kernel: narrowed from the rounded fp32 value
            o[i] = r;
            ob[i] = rb;
        }
        out.store(base + s * 64 * E, o);
        out_b.store(base + s * 64 * E, ob);
    }
}

template <typename T, bool SILU, bool VEC, int NV, int DIR>
__global__ __launch_bounds__(kConvThreads) void conv_bwd_kernel(const vms_conv_bwd_params q) {
    const vms_conv_fwd_params& p = q.f;
    constexpr int E = 16 / sizeof(T);
    constexpr int NW = kConvThreads / 64;
    __shared__ float red[NW][kTaps + 1];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int c = blockIdx.y, b = blockIdx.z;
    const int L = p.seqlen;
    const bool rev = p.reverse != 0;  // logical position t <-> physical L-1-t (anti-causal filter)
    const int base = (blockIdx.x * NW + wave) * (64 * E * NV) + lane * E;
    const auto x = make_row<T, VEC, DIR>(static_cast<const T*>(p.x) + (int64_t)b * p.x_batch_stride + (int64_t)c * p.x_c_stride, L, rev);
    const auto dout = make_row<T, VEC, DIR>(static_cast<const T*>(q.dout) + (int64_t)b * q.dout_batch_stride + (int64_t)c * q.dout_c_stride, L, rev);
    const auto dx = make_row<T, VEC, DIR>(static_cast<T*>(q.dx) + (int64_t)b * q.dx_batch_stride + (int64_t)c * q.dx_c_stride, L, rev);
    float taps[kTaps], bias;
    load_taps(p, c, taps, bias);

    float xv[NV][E + 3], g[NV][E + 3];  // g[s][i] = dout'[l0_s + i], i up to E+2 (right halo)
#pragma unroll
    for (int s = 0; s < NV; ++s) {
        float cur[E], gc[E];
        x.load(base + s * 64 * E, cur);
        dout.load(base + s * 64 * E, gc);
#pragma unroll
        for (int i = 0; i < E; ++i) { xv[s][3 + i] = cur[i]; g[s][i] = gc[i]; }
    }
    float dx_old[NV][E];
    if (q.dx_accumulate) {
#pragma unroll
        for (int s = 0; s < NV; ++s) dx.load(base + s * 64 * E, dx_old[s]);
    }
    // what precedes the wave's first element (lane 0) and what follows its last (lane 63)
    const int lend = base + (NV - 1) * 64 * E + E;
    float edge[3], xr[3], gr[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        edge[j] = x.elem(base - 1 - j, lane == 0);
        xr[j] = x.elem(lend + j, lane == 63);
        gr[j] = dout.elem(lend + j, lane == 63);
    }
#pragma unroll
    for (int s = 0; s < NV; ++s) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float own = xv[s][3 + E - 1 - j];
            const float prev = dpp_mov<DPP_WAVE_SHR1, 0xf>(0.f, own);
            const float left = s == 0 ? edge[j] : lane_value(xv[s - 1][3 + E - 1 - j], 63);
            xv[s][2 - j] = lane == 0 ? left : prev;
        }
        if (SILU) {
#pragma unroll
            for (int i = 0; i < E; ++i) {
                float pre = bias;
#pragma unroll
                for (int k = 0; k < kTaps; ++k) pre = fmaf(taps[k], xv[s][i + k], pre);
                g[s][i] *= silu_grad(pre);
            }
        }
    }
    if (SILU) {  // dout' of the three positions after the wave (gr = 0 in the other lanes)
        float xn[6];
#pragma unroll
        for (int j = 0; j < 3; ++j) { xn[j] = xv[NV - 1][E + j]; xn[3 + j] = xr[j]; }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float pre = bias;
#pragma unroll
            for (int k = 0; k < kTaps; ++k) pre = fmaf(taps[k], xn[j + k], pre);
            gr[j] *= silu_grad(pre);
        }
    }
    // right halo dout'[l0_s+E .. l0_s+E+2]: next lane; lane 63 <- lane 0 of the next strip / the values above
#pragma unroll
    for (int s = 0; s < NV; ++s) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float next = dpp_mov<DPP_WAVE_SHL1, 0xf>(0.f, g[s][j]);
            const float right = s == NV - 1 ? gr[j] : lane_value(g[s + 1][j], 0);
            g[s][E + j] = lane == 63 ? right : next;
        }
    }
    // dx[l] = sum_k taps[k] * dout'[l + 3 - k];  dweight[k] += x[l - 3 + k] * dout'[l];  dbias += dout'[l]
    float dw[kTaps] = {0.f, 0.f, 0.f, 0.f}, db = 0.f;
#pragma unroll
    for (int s = 0; s < NV; ++s) {
        float o[E];
#pragma unroll
        for (int i = 0; i < E; ++i) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < kTaps; ++k) acc = fmaf(taps[k], g[s][i + 3 - k], acc);
            o[i] = acc;
        }
        if (q.dx_accumulate) {  // dx += (vms_hip.h): the other direction's gradient, requested with x and dout above
#pragma unroll
            for (int i = 0; i < E; ++i) o[i] += dx_old[s][i];
        }
        dx.store(base + s * 64 * E, o);
#pragma unroll
        for (int i = 0; i < E; ++i) {
            db += g[s][i];
#pragma unroll
            for (int k = 0; k < kTaps; ++k) dw[k] = fmaf(xv[s][i + k], g[s][i], dw[k]);
        }
    }
    // one wave reduction per tap, the workgroup's waves through LDS, one atomic per (row segment, tap)
#pragma unroll
    for (int k = 0; k < kTaps; ++k) {
        const float t = wave_sum(dw[k]);
        if (lane == 0) red[wave][k] = t;
    }
    {
        const float t = wave_sum(db);
        if (lane == 0) red[wave][kTaps] = t;
    }
    __syncthreads();
    if (threadIdx.x <= kTaps) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += red[w][threadIdx.x];
        const int wi = (int)threadIdx.x - (kTaps - p.width);
        if (threadIdx.x == kTaps) {
            if (q.dbias) atomicAdd(q.dbias + c, t);
        } else if (wi >= 0) {
            atomicAdd(q.dweight + (int64_t)c * q.dweight_c_stride + (int64_t)wi * q.dweight_width_stride, t);
        }
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
            if (q.dx_accumulate) {
                const V old = *reinterpret_cast<const V*>(dx + (int64_t)l * q.dx_l_stride);
#pragma unroll
                for (int e = 0; e < E; ++e) o[e] = static_cast<T>(static_cast<float>(o[e]) + static_cast<float>(old[e]));
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
    VMS_CHECK(p.reverse_from == 0 || (p.reverse_from > 0 && p.reverse_from <= p.batch && p.reverse == 0 && p.x_l_stride == 1 && !p.conv_state),
              "reverse_from must be in (0, batch] with reverse == 0, seqlen-contiguous layout");
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
        // 16-byte rows with whole vectors: buffer-addressed kernels, 4 strips per wave when the row is long
        // buffer-addressed rows whenever a row fits 31-bit byte offsets; `even` = 16-byte aligned rows of whole vectors
        const bool full = (int64_t)p.seqlen * es < ((int64_t)1 << 31);
        const bool even = vec && p.seqlen % E == 0;
        // 4 strips per wave need rows long enough to keep a workgroup's 4 waves busy (4 x 4 x 64 vectors)
        const int nv = full && p.seqlen >= 12 * 64 * E ? 4 : 1;
        const int per_wg = kConvThreads * E * nv;
        dim3 grid((p.seqlen + per_wg - 1) / per_wg, p.dim, p.batch), block(kConvThreads);
#define VMS_K(S_, V_, N_, D_) hipLaunchKernelGGL((conv_fwd_kernel<T, S_, V_, N_, D_>), grid, block, 0, s, p)
#define VMS_D(S_, N_, D_) do { if (even) VMS_K(S_, true, N_, D_); else VMS_K(S_, false, N_, D_); } while (0)
#define VMS_L(S_)                                                                       \
    do {                                                                                \
        if (!full) { if (vec) VMS_K(S_, true, 1, 0); else VMS_K(S_, false, 1, 0); }      \
        else if (nv == 4) { if (p.reverse) VMS_D(S_, 4, 2); else VMS_D(S_, 4, 1); }      \
        else { if (p.reverse) VMS_D(S_, 1, 2); else VMS_D(S_, 1, 1); }                   \
    } while (0)
        if (p.silu_activation) VMS_L(true); else VMS_L(false);
#undef VMS_L
#undef VMS_D
#undef VMS_K
        set_last_kernel(!full ? "conv_fwd_generic" : (nv == 4 ? "conv_fwd_strips4" : "conv_fwd_strips1"));
    } else {
        set_last_kernel("conv_fwd_channel_last");
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
        const bool full = (int64_t)p.seqlen * es < ((int64_t)1 << 31);
        const bool even = vec && p.seqlen % E == 0;
        // 4 strips per wave need rows long enough to keep a workgroup's 4 waves busy (4 x 4 x 64 vectors)
        const int nv = full && p.seqlen >= 12 * 64 * E ? 4 : 1;
        const int per_wg = kConvThreads * E * nv;
        dim3 grid((p.seqlen + per_wg - 1) / per_wg, p.dim, p.batch), block(kConvThreads);
#define VMS_K(S_, V_, N_, D_) hipLaunchKernelGGL((conv_bwd_kernel<T, S_, V_, N_, D_>), grid, block, 0, s, q)
#define VMS_D(S_, N_, D_) do { if (even) VMS_K(S_, true, N_, D_); else VMS_K(S_, false, N_, D_); } while (0)
#define VMS_L(S_)                                                                       \
    do {                                                                                \
        if (!full) { if (vec) VMS_K(S_, true, 1, 0); else VMS_K(S_, false, 1, 0); }      \
        else if (nv == 4) { if (p.reverse) VMS_D(S_, 4, 2); else VMS_D(S_, 4, 1); }      \
        else { if (p.reverse) VMS_D(S_, 1, 2); else VMS_D(S_, 1, 1); }                   \
    } while (0)
        if (p.silu_activation) VMS_L(true); else VMS_L(false);
#undef VMS_L
#undef VMS_D
#undef VMS_K
        set_last_kernel(!full ? "conv_bwd_generic" : (nv == 4 ? "conv_bwd_strips4" : "conv_bwd_strips1"));
    } else {
        set_last_kernel("conv_bwd_channel_last");
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
    if (pp->reverse_from > 0 && pp->reverse_from < pp->batch) {   // ABI v5: the two sub-batches as two launches
        const int rf = pp->reverse_from, es = pp->dtype == VMS_F32 ? 4 : 2;
        vms_conv_fwd_params lo = *pp, hi = *pp;
        lo.batch = rf; lo.reverse_from = 0;
        hi.batch = pp->batch - rf; hi.reverse_from = 0; hi.reverse = 1;
        hi.x = static_cast<const char*>(pp->x) + (int64_t)rf * pp->x_batch_stride * es;
        hi.out = static_cast<char*>(pp->out) + (int64_t)rf * pp->out_batch_stride * es;
        if (int rc = vms_causal_conv1d_fwd(&lo, stream)) return rc;
        return vms_causal_conv1d_fwd(&hi, stream);
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (pp->dtype) {
        case VMS_F32: return conv_fwd_dispatch<float>(*pp, s);
        case VMS_F16: return conv_fwd_dispatch<f16_t>(*pp, s);
        default: return conv_fwd_dispatch<bf16_t>(*pp, s);
    }
}

template <typename T>
static int conv_fwd_dual_dispatch(const vms_conv_fwd_dual_params& q, hipStream_t s) {
    const vms_conv_fwd_params& p = q.f;
    constexpr int E = 16 / sizeof(T);
    const int es = sizeof(T);
    const bool vec = aligned16(p.x) && aligned16(p.out) && aligned16(q.out_b) && mult16(p.x_batch_stride, es) && mult16(p.x_c_stride, es) &&
                     mult16(p.out_batch_stride, es) && mult16(p.out_c_stride, es) && mult16(q.out_b_batch_stride, es) &&
                     mult16(q.out_b_c_stride, es);
    const bool even = vec && p.seqlen % E == 0;
    const int nv = p.seqlen >= 12 * 64 * E ? 4 : 1;
    const int per_wg = kConvThreads * E * nv;
    dim3 grid((p.seqlen + per_wg - 1) / per_wg, p.dim, p.batch), block(kConvThreads);
#define VMS_K(S_, V_, N_) hipLaunchKernelGGL((conv_fwd_dual_kernel<T, S_, V_, N_>), grid, block, 0, s, q)
#define VMS_D(S_, N_) do { if (even) VMS_K(S_, true, N_); else VMS_K(S_, false, N_); } while (0)
#define VMS_L(S_) do { if (nv == 4) VMS_D(S_, 4); else VMS_D(S_, 1); } while (0)
    if (p.silu_activation) VMS_L(true); else VMS_L(false);
#undef VMS_L
#undef VMS_D
#undef VMS_K
    set_last_kernel(nv == 4 ? "conv_fwd_dual_strips4" : "conv_fwd_dual_strips1");
    VMS_LAUNCH_CHECK();
    return VMS_OK;
}

extern "C" int vms_causal_conv1d_fwd_dual(const vms_conv_fwd_dual_params* qq, void* stream) {
    VMS_CHECK(qq != nullptr, "null params");
    const vms_conv_fwd_params& p = qq->f;
    if (int rc = validate_conv(p)) return rc;
    VMS_CHECK(p.out && qq->out_b && qq->weight_b, "out, out_b and weight_b are required");
    VMS_CHECK(p.reverse == 0 && p.reverse_from == 0 && !p.conv_state, "the dual forward takes no direction flags");
    VMS_CHECK(p.x_l_stride == 1 && p.out_l_stride == 1, "x, out and out_b need a unit seqlen stride");
    VMS_CHECK((p.bias == nullptr) == (qq->bias_b == nullptr), "bias_b must be given iff bias is given");
    const int es = p.dtype == VMS_F32 ? 4 : 2;
    if ((int64_t)p.seqlen * es >= ((int64_t)1 << 31)) {   // rows beyond 31-bit byte offsets: the two single-direction launches
        vms_conv_fwd_params lo = p, hi = p;
        hi.reverse = 1;
        hi.weight = qq->weight_b; hi.bias = qq->bias_b; hi.out = qq->out_b;
        hi.weight_c_stride = qq->weight_b_c_stride; hi.weight_width_stride = qq->weight_b_width_stride;
        hi.out_batch_stride = qq->out_b_batch_stride; hi.out_c_stride = qq->out_b_c_stride;
        if (int rc = vms_causal_conv1d_fwd(&lo, stream)) return rc;
        return vms_causal_conv1d_fwd(&hi, stream);
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (p.dtype) {
        case VMS_F32: return conv_fwd_dual_dispatch<float>(*qq, s);
        case VMS_F16: return conv_fwd_dual_dispatch<f16_t>(*qq, s);
        default: return conv_fwd_dual_dispatch<bf16_t>(*qq, s);
    }
}
extern "C" int vms_sizeof_conv_fwd_dual_params(void) { return (int)sizeof(vms_conv_fwd_dual_params); }

extern "C" int vms_causal_conv1d_bwd(const vms_conv_bwd_params* qq, void* stream) {
    VMS_CHECK(qq != nullptr, "null params");
    if (int rc = validate_conv(qq->f)) return rc;
    VMS_CHECK(qq->dout && qq->dx && qq->dweight, "dout, dx and dweight are required");
    VMS_CHECK((qq->f.bias == nullptr) == (qq->dbias == nullptr), "dbias must be given iff bias is given");
    if (qq->f.reverse_from > 0 && qq->f.reverse_from < qq->f.batch) {   // ABI v5: two launches, shared dweight / dbias accumulators
        const int rf = qq->f.reverse_from, es = qq->f.dtype == VMS_F32 ? 4 : 2;
        vms_conv_bwd_params lo = *qq, hi = *qq;
        lo.f.batch = rf; lo.f.reverse_from = 0;
        hi.f.batch = qq->f.batch - rf; hi.f.reverse_from = 0; hi.f.reverse = 1;
        hi.f.x = static_cast<const char*>(qq->f.x) + (int64_t)rf * qq->f.x_batch_stride * es;
        hi.dout = static_cast<const char*>(qq->dout) + (int64_t)rf * qq->dout_batch_stride * es;
        hi.dx = static_cast<char*>(qq->dx) + (int64_t)rf * qq->dx_batch_stride * es;
        if (int rc = vms_causal_conv1d_bwd(&lo, stream)) return rc;
        return vms_causal_conv1d_bwd(&hi, stream);
    }
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
