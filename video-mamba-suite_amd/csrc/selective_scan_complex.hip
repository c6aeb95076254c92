// selective_scan_complex.hip -- selective scan with a COMPLEX A for gfx950 (wave64), forward and backward.
//
// Replaces the weight_t = complex<float> instantiations of the reference (selective_scan.cpp:282-287 / 47;
// selective_scan_fwd_kernel.cuh:223-233, 266-277; selective_scan_bwd_kernel.cuh:330-436; SSMScanOp<float4>,
// selective_scan_common.h:117-128).  Semantics = selective_scan_ref's complex branch (SSI:111-116, 144-145):
//     a_l = exp(delta_l A_n),  x_l = a_l x_{l-1} + delta_l u_l B_{n,l},  y_l = 2 Re(sum_n C_{n,l} x_{l,n}) + D u_l
// with gradients of complex parameters in PyTorch's convention (dL/dRe + i dL/dIm).  No suite model uses a complex A
// (S4D-real initialisation only, mamba_simple.py:112-118): this is the completeness path behind selective_scan_fn, built like
// the generic real kernels (selective_scan_fwd.hip / selective_scan_bwd.hip) and not tuned further:
//   * one wave = one (batch, dim) row, 64 * 8-element chunks, lane j owns 8 consecutive elements;
//   * per state: in-register scan of the lane's elements -> DPP scan of the (a, x) lane aggregates with the complex
//     monoid (a1 a0, a1 x0 + x1) -> seeded second pass; the adjoint g_l = 2 dy_l conj(C_l) + conj(a_{l+1}) g_{l+1}
//     the same way from the right;
//   * layouts (vms_hip.h, is_complex): A, constant B / C and their gradients are (re, im) float pairs with strides in
//     complex elements; variable B / C are the reference's real (batch, groups, dstate, 2 seqlen) tensors of interleaved
//     pairs (their gradients fp32 of the same shape: the products of a backward workgroup's 8 rows are summed in LDS, then
//     one atomic per value and workgroup);
//     x is complex (batch, dim, n_chunks, 2 dstate) with the slots of the real kernels, optionally followed by the
//     state after every 512 elements (x_has_sub == 1: the backward's chunk seeds).
#include "vms_common.h"

namespace vms {

namespace {

constexpr int kCK = 8;               // elements per lane
constexpr int kCCS = kWave * kCK;    // elements per wave chunk (512)
constexpr int kCRows = 4;            // rows (waves) per forward workgroup
constexpr int kCBRows = 8;           // rows (waves) per backward workgroup: their dB / dC products are summed in LDS first

struct cf {
    float re, im;
};
__device__ __forceinline__ cf cmul(cf a, cf b) { return cf{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cf cfma(cf a, cf b, cf c) {   // a b + c
    return cf{fmaf(a.re, b.re, fmaf(-a.im, b.im, c.re)), fmaf(a.re, b.im, fmaf(a.im, b.re, c.im))};
}
__device__ __forceinline__ cf cconj(cf a) { return cf{a.re, -a.im}; }
// exp(dl (Ar + i Ai)): the magnitude with one v_exp_f32, the phase with the accurate sincosf (arguments are not bounded)
__device__ __forceinline__ cf cexp_scaled(float dl, cf A) {
    const float m = fast_exp2(dl * A.re * kLog2e);
    float s, c;
    sincosf(dl * A.im, &s, &c);
    return cf{m * c, m * s};
}

// inclusive scan of the complex monoid over the 64 lanes in lane order (the real one: vms_common.h wave_scan_inclusive)
__device__ __forceinline__ void cwave_scan_inclusive(cf& a, cf& x) {
#define VMS_CSTEP(C, M)                                                        \
    {                                                                          \
        const cf xp = cf{dpp_mov<C, M>(0.f, x.re), dpp_mov<C, M>(0.f, x.im)};  \
        const cf ap = cf{dpp_mov<C, M>(1.f, a.re), dpp_mov<C, M>(0.f, a.im)};  \
        x = cfma(a, xp, x);                                                    \
        a = cmul(a, ap);                                                       \
    }
    VMS_CSTEP(DPP_ROW_SHR1, 0xf)
    VMS_CSTEP(DPP_ROW_SHR2, 0xf)
    VMS_CSTEP(DPP_ROW_SHR4, 0xf)
    VMS_CSTEP(DPP_ROW_SHR8, 0xf)
    VMS_CSTEP(DPP_ROW_BCAST15, 0xa)
    VMS_CSTEP(DPP_ROW_BCAST31, 0xc)
#undef VMS_CSTEP
}

// the same monoid from the last lane towards lane 0 (vms_common.h wave_scan_inclusive_reverse)
__device__ __forceinline__ void cwave_scan_inclusive_reverse(cf& a, cf& x) {
#define VMS_CRSTEP(C)                                                              \
    {                                                                              \
        const cf xp = cf{dpp_mov<C, 0xf>(0.f, x.re), dpp_mov<C, 0xf>(0.f, x.im)};  \
        const cf ap = cf{dpp_mov<C, 0xf>(1.f, a.re), dpp_mov<C, 0xf>(0.f, a.im)};  \
        x = cfma(a, xp, x);                                                        \
        a = cmul(a, ap);                                                           \
    }
    VMS_CRSTEP(DPP_ROW_SHL1)
    VMS_CRSTEP(DPP_ROW_SHL2)
    VMS_CRSTEP(DPP_ROW_SHL4)
    VMS_CRSTEP(DPP_ROW_SHL8)
#undef VMS_CRSTEP
    auto rl = [](cf v, int l) { return cf{readlane_f(v.re, l), readlane_f(v.im, l)}; };
    const cf a1 = rl(a, 16), x1 = rl(x, 16), a2 = rl(a, 32), x2 = rl(x, 32), a3 = rl(a, 48), x3 = rl(x, 48);
    const cf a23 = cmul(a2, a3), x23 = cfma(a2, x3, x2);
    const cf a123 = cmul(a1, a23), x123 = cfma(a1, x23, x1);
    const int row = (threadIdx.x & 63) >> 4;
    const cf sa = row == 0 ? a123 : (row == 1 ? a23 : (row == 2 ? a3 : cf{1.f, 0.f}));
    const cf sx = row == 0 ? x123 : (row == 1 ? x23 : (row == 2 ? x3 : cf{0.f, 0.f}));
    x = cfma(a, sx, x);
    a = cmul(a, sa);
}

__device__ __forceinline__ cf cshift_right(cf idv, cf v) {   // lane i <- lane i - 1, lane 0 <- idv
    return cf{dpp_mov<DPP_WAVE_SHR1, 0xf>(idv.re, v.re), dpp_mov<DPP_WAVE_SHR1, 0xf>(idv.im, v.im)};
}
__device__ __forceinline__ cf cshift_left(cf idv, cf v) {    // lane i <- lane i + 1, lane 63 <- idv
    return cf{dpp_mov<DPP_WAVE_SHL1, 0xf>(idv.re, v.re), dpp_mov<DPP_WAVE_SHL1, 0xf>(idv.im, v.im)};
}

// the K logical positions [l0, l0 + K) of a row of L interleaved (re, im) pairs; positions past the end read 0
template <typename T, int K, bool VEC>
__device__ __forceinline__ void load_pairs(const T* __restrict__ row, int l0, int L, bool rev, cf (&out)[K]) {
    if (!rev) {
        float t[2 * K];
        load_blocked<T, 2 * K, VEC>(row + 2 * (int64_t)l0, 2 * (L - l0), t);
#pragma unroll
        for (int i = 0; i < K; ++i) out[i] = cf{t[2 * i], t[2 * i + 1]};
    } else {
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int64_t ph = L - 1 - l0 - i;
            out[i] = l0 + i < L ? cf{static_cast<float>(row[2 * ph]), static_cast<float>(row[2 * ph + 1])} : cf{0.f, 0.f};
        }
    }
}

__device__ __forceinline__ cf ldc(const float* p, int64_t i) { return cf{p[2 * i], p[2 * i + 1]}; }

template <typename T, bool VB, bool VC, bool HZ, bool VEC>
__global__ __launch_bounds__(kCRows* kWave) void cscan_fwd_kernel(const vms_scan_fwd_params p) {
    extern __shared__ float smem[];
    constexpr int K = kCK, CS = kCCS;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tiles = (p.dim + kCRows - 1) / kCRows;
    const int b = blockIdx.x / tiles;
    const int d = (blockIdx.x - b * tiles) * kCRows + wave;
    if (d >= p.dim) return;
    const int g = d / (p.dim / p.n_groups);
    const int L = p.seqlen, N = p.dstate;
    const bool rev = p.reverse != 0 || (p.reverse_from > 0 && b >= p.reverse_from);   // per batch entry (workgroup-uniform)
    volatile lds_f32* h = (lds_f32*)smem + wave * 2 * N;   // running complex state per n (wave-private)
    for (int n = lane; n < 2 * N; n += kWave) h[n] = 0.f;
    const int64_t xpitch = 2 * (p.x_chunk_stride ? p.x_chunk_stride : 2 * (int64_t)N);   // floats

    const T* u = static_cast<const T*>(p.u) + (int64_t)b * p.u_batch_stride + (int64_t)d * p.u_d_stride;
    const T* dt = static_cast<const T*>(p.delta) + (int64_t)b * p.delta_batch_stride + (int64_t)d * p.delta_d_stride;
    T* out = static_cast<T*>(p.out) + (int64_t)b * p.out_batch_stride + (int64_t)d * p.out_d_stride;
    const T* z = HZ ? static_cast<const T*>(p.z) + (int64_t)b * p.z_batch_stride + (int64_t)d * p.z_d_stride : nullptr;
    T* out_z = HZ ? static_cast<T*>(p.out_z) + (int64_t)b * p.out_z_batch_stride + (int64_t)d * p.out_z_d_stride : nullptr;
    const float* A = static_cast<const float*>(p.A) + 2 * (int64_t)d * p.A_d_stride;
    const T* Bv = VB ? static_cast<const T*>(p.B) + (int64_t)b * p.B_batch_stride + (int64_t)g * p.B_group_stride : nullptr;
    const T* Cv = VC ? static_cast<const T*>(p.C) + (int64_t)b * p.C_batch_stride + (int64_t)g * p.C_group_stride : nullptr;
    const float* Bc = !VB ? static_cast<const float*>(p.B) + 2 * (int64_t)d * p.B_d_stride : nullptr;
    const float* Cc = !VC ? static_cast<const float*>(p.C) + 2 * (int64_t)d * p.C_d_stride : nullptr;
    float* xck = static_cast<float*>(p.x) + ((int64_t)b * p.dim + d) * p.n_chunks * xpitch;
    const float Dd = p.D ? static_cast<const float*>(p.D)[d] : 0.f;
    const float bias = p.delta_bias ? static_cast<const float*>(p.delta_bias)[d] : 0.f;

    const int n_kchunks = (L + CS - 1) / CS;
    for (int c = 0; c < n_kchunks; ++c) {
        const int l0 = c * CS + lane * K;
        const int nv = L - l0;
        float uv[K], dl[K], du[K], y[K];
        load_dir<T, K, VEC>(u, l0, L, rev, uv);
        load_dir<T, K, VEC>(dt, l0, L, rev, dl);
#pragma unroll
        for (int i = 0; i < K; ++i) {
            float t = dl[i] + bias;
            if (p.delta_softplus) t = softplusf_(t);
            dl[i] = i < nv ? t : 0.f;   // past the end: the identity (a = 1, b = 0)
            du[i] = dl[i] * uv[i];
            y[i] = Dd * uv[i];
        }
        for (int n = 0; n < N; ++n) {
            const cf An = ldc(A, n * p.A_dstate_stride);
            cf Bn[K], Cn[K];
            if (VB) load_pairs<T, K, VEC>(Bv + (int64_t)n * p.B_dstate_stride, l0, L, rev, Bn);
            if (VC) load_pairs<T, K, VEC>(Cv + (int64_t)n * p.C_dstate_stride, l0, L, rev, Cn);
            const cf bconst = VB ? cf{1.f, 0.f} : ldc(Bc, n * p.B_dstate_stride);
            const cf cconst = VC ? cf{1.f, 0.f} : ldc(Cc, n * p.C_dstate_stride);
            cf a[K], bx[K];
            cf pa = cf{1.f, 0.f}, px = cf{0.f, 0.f};
#pragma unroll
            for (int i = 0; i < K; ++i) {
                a[i] = cexp_scaled(dl[i], An);
                const cf Bi = VB ? Bn[i] : bconst;
                bx[i] = cf{du[i] * Bi.re, du[i] * Bi.im};
                px = cfma(a[i], px, bx[i]);
                pa = cmul(pa, a[i]);
            }
            cwave_scan_inclusive(pa, px);
            const cf ea = cshift_right(cf{1.f, 0.f}, pa), ex = cshift_right(cf{0.f, 0.f}, px);
            const cf hin = cf{h[2 * n], h[2 * n + 1]};
            cf xs = cfma(ea, hin, ex);
            const cf hout = cfma(cf{readlane_f(pa.re, 63), readlane_f(pa.im, 63)}, hin,
                                 cf{readlane_f(px.re, 63), readlane_f(px.im, 63)});
            if (lane == 0) {
                h[2 * n] = hout.re;
                h[2 * n + 1] = hout.im;
            }
#pragma unroll
            for (int i = 0; i < K; ++i) {
                xs = cfma(a[i], xs, bx[i]);
                const cf Ci = VC ? Cn[i] : cconst;
                y[i] += 2.f * (Ci.re * xs.re - Ci.im * xs.im);
            }
        }
        store_dir<T, K, VEC>(out, l0, L, rev, y);
        if (HZ) {
            float zv[K];
            load_dir<T, K, VEC>(z, l0, L, rev, zv);
#pragma unroll
            for (int i = 0; i < K; ++i) y[i] *= zv[i] * sigmoidf_(zv[i]);
            if (p.out_z_accumulate) {
                float old[K];
                load_dir<T, K, VEC>(out_z, l0, L, rev, old);
#pragma unroll
                for (int i = 0; i < K; ++i) y[i] += old[i];
            }
            store_dir<T, K, VEC>(out_z, l0, L, rev, y);
        }
        // the state after every 512 elements (x_has_sub == 1), behind the reference-shaped slots
        if (p.x_has_sub == 1) {
            float* xs_ = xck + (int64_t)(c >> 2) * xpitch + 4 * N + (c & 3) * 2 * N;
            for (int n = lane; n < 2 * N; n += kWave) xs_[n] = h[n];
        }
        // the reference-shaped slots: [2n] after the first 1024 elements of a 2048-chunk, [2n + 1] at its end
        const int pos = (c + 1) * CS;
        const bool last = c == n_kchunks - 1;
        if ((pos & 1023) == 0 || last) {
            const int blk = last ? (L - 1) / 2048 : (pos - 1) / 2048;
            const int r = (last ? L : pos) - blk * 2048;
            float* xb = xck + (int64_t)blk * xpitch;
            const bool w_even = r <= 1024, w_odd = r == 2048 || last;
            for (int n = lane; n < 2 * N; n += kWave) {
                const float s = h[n];
                const int st = n >> 1, ri = n & 1;
                if (w_even) xb[(2 * st) * 2 + ri] = s;
                if (w_odd) xb[(2 * st + 1) * 2 + ri] = s;
            }
        }
    }
}

template <typename T, bool VB, bool VC, bool HZ, bool VEC>
__global__ __launch_bounds__(kCBRows* kWave) void cscan_bwd_kernel(const vms_scan_bwd_params q) {
    const vms_scan_fwd_params& p = q.f;
    extern __shared__ float smem[];
    constexpr int K = kCK, CS = kCCS;
    constexpr int kSlab = 2 * kCBRows * K * kWave * 2;   // floats: [tensor][wave][element][lane] (re, im) pairs = 64 KB
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tiles = (p.dim + kCBRows - 1) / kCBRows;
    const int b = blockIdx.x / tiles;
    const int d0 = (blockIdx.x - b * tiles) * kCBRows;
    const bool row_ok = d0 + wave < p.dim;      // rows past the end read nothing, contribute zeros and meet the barriers
    const int d = row_ok ? d0 + wave : p.dim - 1;
    const int dpg = p.dim / p.n_groups;
    const int g = d / dpg;
    // variable dB / dC are sums over the rows of a group: when the workgroup's rows share one (the usual case) their products are
    // summed in LDS (plain stores, one slot per wave) and leave as ONE pair of atomics per position and workgroup -- one atomic per
    // (row, state, value) as the reference issues them met 1024-way contention per address here: 189 ms at (8, 1024, 8192)
    const int d_last = (d0 + kCBRows < p.dim ? d0 + kCBRows : p.dim) - 1;
    const bool same_group = (d0 / dpg) == (d_last / dpg);
    const int Lr = row_ok ? p.seqlen : 0;
    const int L = p.seqlen, N = p.dstate;
    const bool rev = p.reverse != 0 || (p.reverse_from > 0 && b >= p.reverse_from);   // per batch entry (workgroup-uniform)
    typedef float v2f __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) v2f lds_c2;
    lds_c2* const slab = (lds_c2*)smem;
    // per wave: adjoint entering from the right, a of the first element of the chunk to the right, dA / constant dB / dC
    volatile lds_f32* wv = (lds_f32*)smem + kSlab + wave * (10 * N);
    volatile lds_f32 *gcarry = wv, *anext = wv + 2 * N, *dA_acc = wv + 4 * N, *dBc_acc = wv + 6 * N, *dCc_acc = wv + 8 * N;
    for (int n = lane; n < 2 * N; n += kWave) {
        gcarry[n] = 0.f;
        anext[n] = (n & 1) ? 0.f : 1.f;
        dA_acc[n] = 0.f;
        dBc_acc[n] = 0.f;
        dCc_acc[n] = 0.f;
    }
    const T* u = static_cast<const T*>(p.u) + (int64_t)b * p.u_batch_stride + (int64_t)d * p.u_d_stride;
    const T* dt = static_cast<const T*>(p.delta) + (int64_t)b * p.delta_batch_stride + (int64_t)d * p.delta_d_stride;
    const T* dout = static_cast<const T*>(q.dout) + (int64_t)b * q.dout_batch_stride + (int64_t)d * q.dout_d_stride;
    T* du = static_cast<T*>(q.du) + (int64_t)b * q.du_batch_stride + (int64_t)d * q.du_d_stride;
    T* ddelta = static_cast<T*>(q.ddelta) + (int64_t)b * q.ddelta_batch_stride + (int64_t)d * q.ddelta_d_stride;
    const T* z = HZ ? static_cast<const T*>(p.z) + (int64_t)b * p.z_batch_stride + (int64_t)d * p.z_d_stride : nullptr;
    const T* outp = HZ ? static_cast<const T*>(p.out) + (int64_t)b * p.out_batch_stride + (int64_t)d * p.out_d_stride : nullptr;
    T* dz = HZ ? static_cast<T*>(q.dz) + (int64_t)b * q.dz_batch_stride + (int64_t)d * q.dz_d_stride : nullptr;
    T* out_z = (HZ && p.out_z) ? static_cast<T*>(p.out_z) + (int64_t)b * p.out_z_batch_stride + (int64_t)d * p.out_z_d_stride : nullptr;
    const float* A = static_cast<const float*>(p.A) + 2 * (int64_t)d * p.A_d_stride;
    const T* Bv = VB ? static_cast<const T*>(p.B) + (int64_t)b * p.B_batch_stride + (int64_t)g * p.B_group_stride : nullptr;
    const T* Cv = VC ? static_cast<const T*>(p.C) + (int64_t)b * p.C_batch_stride + (int64_t)g * p.C_group_stride : nullptr;
    const float* Bc = !VB ? static_cast<const float*>(p.B) + 2 * (int64_t)d * p.B_d_stride : nullptr;
    const float* Cc = !VC ? static_cast<const float*>(p.C) + 2 * (int64_t)d * p.C_d_stride : nullptr;
    const int64_t xpitch = 2 * (p.x_chunk_stride ? p.x_chunk_stride : 2 * (int64_t)N);
    const float* xck = p.x ? static_cast<const float*>(p.x) + ((int64_t)b * p.dim + d) * p.n_chunks * xpitch : nullptr;
    const float Dd = p.D ? static_cast<const float*>(p.D)[d] : 0.f;
    const float bias = p.delta_bias ? static_cast<const float*>(p.delta_bias)[d] : 0.f;
    float* dBg = VB ? q.dB + (int64_t)b * q.dB_batch_stride + (int64_t)g * q.dB_group_stride : nullptr;
    float* dCg = VC ? q.dC + (int64_t)b * q.dC_batch_stride + (int64_t)g * q.dC_group_stride : nullptr;

    float dD_acc = 0.f, dbias_acc = 0.f;
    const int n_kchunks = (L + CS - 1) / CS;
    for (int c = n_kchunks - 1; c >= 0; --c) {
        const int l0 = c * CS + lane * K;
        const int nv = Lr - l0;
        float uv[K], dl[K], dy[K], duv[K], ddl[K];
        load_dir<T, K, VEC>(u, l0, Lr, rev, uv);
        load_dir<T, K, VEC>(dt, l0, Lr, rev, dl);
        load_dir<T, K, VEC>(dout, l0, Lr, rev, dy);
#pragma unroll
        for (int i = 0; i < K; ++i) {
            float t = dl[i] + bias;
            if (p.delta_softplus) t = softplusf_(t);
            dl[i] = i < nv ? t : 0.f;
        }
        if (HZ) {
            float zv[K], ov[K], dzv[K];
            load_dir<T, K, VEC>(z, l0, Lr, rev, zv);
            load_dir<T, K, VEC>(outp, l0, Lr, rev, ov);
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const float s = sigmoidf_(zv[i]);
                const float silu = zv[i] * s;
                dzv[i] = dy[i] * ov[i] * s * (1.f + zv[i] * (1.f - s));
                dy[i] *= silu;
                ov[i] *= silu;
            }
            if (q.dz_accumulate) {
                float old[K];
                load_dir<T, K, VEC>(dz, l0, Lr, rev, old);
#pragma unroll
                for (int i = 0; i < K; ++i) dzv[i] += old[i];
            }
            store_dir<T, K, VEC>(dz, l0, Lr, rev, dzv);
            if (out_z) store_dir<T, K, VEC>(out_z, l0, Lr, rev, ov);
        }
#pragma unroll
        for (int i = 0; i < K; ++i) {
            duv[i] = Dd * dy[i];
            dD_acc = fmaf(dy[i], uv[i], dD_acc);
            ddl[i] = 0.f;
        }
        // the state entering this chunk: the forward's 512-element checkpoint c - 1
        const float* xin = c > 0 ? xck + (int64_t)((c - 1) >> 2) * xpitch + 4 * N + ((c - 1) & 3) * 2 * N : nullptr;
        for (int n = 0; n < N; ++n) {
            const cf An = ldc(A, n * p.A_dstate_stride);
            cf Bn[K], Cn[K];
            if (VB) load_pairs<T, K, VEC>(Bv + (int64_t)n * p.B_dstate_stride, l0, Lr, rev, Bn);
            if (VC) load_pairs<T, K, VEC>(Cv + (int64_t)n * p.C_dstate_stride, l0, Lr, rev, Cn);
            const cf bconst = VB ? cf{1.f, 0.f} : ldc(Bc, n * p.B_dstate_stride);
            const cf cconst = VC ? cf{1.f, 0.f} : ldc(Cc, n * p.C_dstate_stride);
            // ---- forward re-scan: x_i of the lane's elements ----
            cf a[K], xs[K];
            cf pa = cf{1.f, 0.f}, px = cf{0.f, 0.f};
#pragma unroll
            for (int i = 0; i < K; ++i) {
                a[i] = cexp_scaled(dl[i], An);
                const cf Bi = VB ? Bn[i] : bconst;
                const float dlu = dl[i] * uv[i];
                px = cfma(a[i], px, cf{dlu * Bi.re, dlu * Bi.im});
                pa = cmul(pa, a[i]);
            }
            cwave_scan_inclusive(pa, px);
            const cf ea = cshift_right(cf{1.f, 0.f}, pa), ex = cshift_right(cf{0.f, 0.f}, px);
            const cf hin = xin ? cf{xin[2 * n], xin[2 * n + 1]} : cf{0.f, 0.f};
            cf xrun = cfma(ea, hin, ex);
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const cf Bi = VB ? Bn[i] : bconst;
                const float dlu = dl[i] * uv[i];
                xrun = cfma(a[i], xrun, cf{dlu * Bi.re, dlu * Bi.im});
                xs[i] = xrun;
            }
            // ---- adjoint, right to left: g_i = 2 dy_i conj(C_i) + conj(a_{i+1}) g_{i+1} ----
            const cf a_right = cconj(cshift_left(cf{anext[2 * n], anext[2 * n + 1]}, a[0]));   // lane 63 <- the chunk to the right
            cf ra = cf{1.f, 0.f}, rg = cf{0.f, 0.f};
#pragma unroll
            for (int i = K - 1; i >= 0; --i) {
                const cf alpha = i == K - 1 ? a_right : cconj(a[i + 1]);
                const cf Ci = VC ? Cn[i] : cconst;
                rg = cfma(alpha, rg, cf{2.f * dy[i] * Ci.re, -2.f * dy[i] * Ci.im});
                ra = cmul(ra, alpha);
            }
            cwave_scan_inclusive_reverse(ra, rg);
            const cf esa = cshift_left(cf{1.f, 0.f}, ra), esx = cshift_left(cf{0.f, 0.f}, rg);
            const cf gin = cf{gcarry[2 * n], gcarry[2 * n + 1]};
            cf grun = cfma(esa, gin, esx);
            const cf gout = cfma(cf{readlane_f(ra.re, 0), readlane_f(ra.im, 0)}, gin, cf{readlane_f(rg.re, 0), readlane_f(rg.im, 0)});
            const cf a_first = cf{readlane_f(a[0].re, 0), readlane_f(a[0].im, 0)};
            if (lane == 0) {
                gcarry[2 * n] = gout.re;
                gcarry[2 * n + 1] = gout.im;
                anext[2 * n] = a_first.re;
                anext[2 * n + 1] = a_first.im;
            }
            cf dA_loc = cf{0.f, 0.f}, dBc_loc = cf{0.f, 0.f}, dCc_loc = cf{0.f, 0.f};
#pragma unroll
            for (int i = K - 1; i >= 0; --i) {
                const cf alpha = i == K - 1 ? a_right : cconj(a[i + 1]);
                const cf Ci = VC ? Cn[i] : cconst;
                const cf Bi = VB ? Bn[i] : bconst;
                grun = cfma(alpha, grun, cf{2.f * dy[i] * Ci.re, -2.f * dy[i] * Ci.im});
                const cf gx = grun;
                const float dlu = dl[i] * uv[i];
                const cf ax = cf{xs[i].re - dlu * Bi.re, xs[i].im - dlu * Bi.im};   // a_i x_{i-1}
                const float bg = Bi.re * gx.re + Bi.im * gx.im;                      // Re(conj(B) g)
                const cf Aax = cmul(An, ax);
                duv[i] = fmaf(dl[i], bg, duv[i]);
                ddl[i] = fmaf(uv[i], bg, ddl[i]) + (Aax.re * gx.re + Aax.im * gx.im);
                const cf cag = cmul(cconj(ax), gx);
                dA_loc.re = fmaf(dl[i], cag.re, dA_loc.re);
                dA_loc.im = fmaf(dl[i], cag.im, dA_loc.im);
                const cf dBi = cf{dlu * gx.re, dlu * gx.im};
                const cf dCi = cf{2.f * dy[i] * xs[i].re, -2.f * dy[i] * xs[i].im};
                const int64_t ph = rev ? L - 1 - (l0 + i) : l0 + i;
                if (VB) {
                    if (same_group) {
                        slab[((0 * kCBRows + wave) * K + i) * kWave + lane] = i < nv ? v2f{dBi.re, dBi.im} : v2f{0.f, 0.f};
                    } else if (i < nv) {
                        float* t = dBg + (int64_t)n * q.dB_dstate_stride + 2 * ph;
                        atomicAdd(t, dBi.re);
                        atomicAdd(t + 1, dBi.im);
                    }
                } else {
                    dBc_loc.re += dBi.re;
                    dBc_loc.im += dBi.im;
                }
                if (VC) {
                    if (same_group) {
                        slab[((1 * kCBRows + wave) * K + i) * kWave + lane] = i < nv ? v2f{dCi.re, dCi.im} : v2f{0.f, 0.f};
                    } else if (i < nv) {
                        float* t = dCg + (int64_t)n * q.dC_dstate_stride + 2 * ph;
                        atomicAdd(t, dCi.re);
                        atomicAdd(t + 1, dCi.im);
                    }
                } else {
                    dCc_loc.re += dCi.re;
                    dCc_loc.im += dCi.im;
                }
            }
            {
                const float tr = wave_sum(dA_loc.re), ti = wave_sum(dA_loc.im);
                if (lane == 0) {
                    dA_acc[2 * n] += tr;
                    dA_acc[2 * n + 1] += ti;
                }
            }
            if (!VB) {
                const float tr = wave_sum(dBc_loc.re), ti = wave_sum(dBc_loc.im);
                if (lane == 0) {
                    dBc_acc[2 * n] += tr;
                    dBc_acc[2 * n + 1] += ti;
                }
            }
            if (!VC) {
                const float tr = wave_sum(dCc_loc.re), ti = wave_sum(dCc_loc.im);
                if (lane == 0) {
                    dCc_acc[2 * n] += tr;
                    dCc_acc[2 * n + 1] += ti;
                }
            }
            if ((VB || VC) && same_group) {   // workgroup-uniform: every wave walks the same chunks and states
                __syncthreads();
                const int j = threadIdx.x;    // position j of the chunk = element j % K of lane j / K (blockDim == CS)
                if (c * CS + j < L) {
                    const int64_t ph = rev ? L - 1 - (c * CS + j) : c * CS + j;
                    const int src = (j % K) * kWave + j / K;
#pragma unroll
                    for (int ten = 0; ten < 2; ++ten) {
                        if (ten == 0 ? !VB : !VC) continue;
                        v2f acc = v2f{0.f, 0.f};
#pragma unroll
                        for (int w = 0; w < kCBRows; ++w) {
                            acc += slab[(ten * kCBRows + w) * K * kWave + src];
                        }
                        float* t = (ten == 0 ? q.dB + (int64_t)b * q.dB_batch_stride + (int64_t)(d0 / dpg) * q.dB_group_stride + (int64_t)n * q.dB_dstate_stride
                                             : q.dC + (int64_t)b * q.dC_batch_stride + (int64_t)(d0 / dpg) * q.dC_group_stride + (int64_t)n * q.dC_dstate_stride) + 2 * ph;
                        atomicAdd(t, acc.x);
                        atomicAdd(t + 1, acc.y);
                    }
                }
                __syncthreads();
            }
        }
        // softplus chain (selective_scan_bwd_kernel.cuh:439-452) and stores
        {
            float raw[K];
            load_dir<T, K, VEC>(dt, l0, Lr, rev, raw);
#pragma unroll
            for (int i = 0; i < K; ++i) {
                if (p.delta_softplus) {
                    const float r = raw[i] + bias;
                    ddl[i] = r <= 20.f ? ddl[i] * sigmoidf_(r) : ddl[i];
                }
                if (i < nv) dbias_acc += ddl[i];
            }
        }
        store_dir<T, K, VEC>(du, l0, Lr, rev, duv);
        store_dir<T, K, VEC>(ddelta, l0, Lr, rev, ddl);
    }
    if (q.dD) {
        const float t = wave_sum(dD_acc);
        if (lane == 0 && row_ok) atomicAdd(q.dD + d, t);
    }
    if (q.ddelta_bias) {
        const float t = wave_sum(dbias_acc);
        if (lane == 0 && row_ok) atomicAdd(q.ddelta_bias + d, t);
    }
    for (int n = lane; n < 2 * N && row_ok; n += kWave) {
        const int st = n >> 1, ri = n & 1;
        atomicAdd(q.dA + 2 * ((int64_t)d * q.dA_d_stride + (int64_t)st * q.dA_dstate_stride) + ri, dA_acc[n]);
        if (!VB) atomicAdd(q.dB + 2 * ((int64_t)d * q.dB_d_stride + (int64_t)st * q.dB_dstate_stride) + ri, dBc_acc[n]);
        if (!VC) atomicAdd(q.dC + 2 * ((int64_t)d * q.dC_d_stride + (int64_t)st * q.dC_dstate_stride) + ri, dCc_acc[n]);
    }
}

template <typename T, bool VB, bool VC, bool HZ>
int launch_cfwd(const vms_scan_fwd_params& p, bool vec, hipStream_t stream) {
    const int tiles = (p.dim + kCRows - 1) / kCRows;
    dim3 grid(p.batch * tiles), block(kCRows * kWave);
    const size_t smem = sizeof(float) * kCRows * 2 * p.dstate;
    if (vec)
        hipLaunchKernelGGL((cscan_fwd_kernel<T, VB, VC, HZ, true>), grid, block, smem, stream, p);
    else
        hipLaunchKernelGGL((cscan_fwd_kernel<T, VB, VC, HZ, false>), grid, block, smem, stream, p);
    VMS_LAUNCH_CHECK();
    return VMS_OK;
}

template <typename T, bool VB, bool VC, bool HZ>
int launch_cbwd(const vms_scan_bwd_params& q, bool vec, hipStream_t stream) {
    const vms_scan_fwd_params& p = q.f;
    const int tiles = (p.dim + kCBRows - 1) / kCBRows;
    dim3 grid(p.batch * tiles), block(kCBRows * kWave);
    static_assert(kCBRows * kWave == kCCS, "the slab flush maps one thread to one position of a chunk");
    const size_t smem = sizeof(float) * (2 * kCBRows * kCK * kWave * 2 + kCBRows * 10 * p.dstate);   // 64 KB slab + per-wave state arrays
    // more than 64 KB of dynamic LDS needs the attribute on each device the kernel runs on: set once per (instantiation, device)
    // to what the largest dstate (256, validate_scan_common) can ask for, like the real-A kernels (vms_hip.h "Conventions")
    static PerDeviceOnce attr_once;
    const hipError_t arc = attr_once.run([&]() -> hipError_t {
        const int smem_max = (int)(sizeof(float) * (2 * kCBRows * kCK * kWave * 2 + kCBRows * 10 * 256));
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&cscan_bwd_kernel<T, VB, VC, HZ, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem_max);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&cscan_bwd_kernel<T, VB, VC, HZ, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, smem_max);
        return e;
    });
    if (arc != hipSuccess) {
        set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed: %s", hipGetErrorString(arc));
        return VMS_ERR_LAUNCH;
    }
    if (vec)
        hipLaunchKernelGGL((cscan_bwd_kernel<T, VB, VC, HZ, true>), grid, block, smem, stream, q);
    else
        hipLaunchKernelGGL((cscan_bwd_kernel<T, VB, VC, HZ, false>), grid, block, smem, stream, q);
    VMS_LAUNCH_CHECK();
    return VMS_OK;
}

#define VMS_CCASES(FN, ARG)                                       \
    VMS_CCASE(FN, ARG, true, true, true)                          \
    VMS_CCASE(FN, ARG, true, true, false)                         \
    VMS_CCASE(FN, ARG, true, false, true)                         \
    VMS_CCASE(FN, ARG, true, false, false)                        \
    VMS_CCASE(FN, ARG, false, true, true)                         \
    VMS_CCASE(FN, ARG, false, true, false)                        \
    VMS_CCASE(FN, ARG, false, false, true)                        \
    VMS_CCASE(FN, ARG, false, false, false)
#define VMS_CCASE(FN, ARG, B_, C_, Z_) \
    if (vb == B_ && vc == C_ && hz == Z_) return FN<T, B_, C_, Z_>(ARG, vec, s);

template <typename T>
int dispatch_cfwd(const vms_scan_fwd_params& p, bool vec, hipStream_t s) {
    const bool vb = p.is_variable_B, vc = p.is_variable_C, hz = p.z != nullptr;
    VMS_CCASES(launch_cfwd, p)
    return VMS_ERR_INVALID_ARG;
}
template <typename T>
int dispatch_cbwd(const vms_scan_bwd_params& q, bool vec, hipStream_t s) {
    const bool vb = q.f.is_variable_B, vc = q.f.is_variable_C, hz = q.f.z != nullptr;
    VMS_CCASES(launch_cbwd, q)
    return VMS_ERR_INVALID_ARG;
}
#undef VMS_CCASE
#undef VMS_CCASES

int validate_complex(const vms_scan_fwd_params& p) {
    VMS_CHECK(p.x_has_sub == 0 || p.x_has_sub == 1, "complex A: x_has_sub must be 0 or 1");
    VMS_CHECK(p.x_has_sub == 0 || p.x_chunk_stride >= 6 * (int64_t)p.dstate,
              "complex A: x_has_sub == 1 needs an x pitch >= 6 * dstate complex elements");
    VMS_CHECK(p.x_chunk_stride == 0 || p.x_chunk_stride >= 2 * (int64_t)p.dstate, "x pitch < 2 * dstate");
    return VMS_OK;
}

}  // namespace

int launch_scan_fwd_complex(const vms_scan_fwd_params& p, bool vec, hipStream_t s) {
    if (int rc = validate_complex(p)) return rc;
    set_last_kernel("scan_fwd_complex");
    switch (p.dtype) {
        case VMS_F32: return dispatch_cfwd<float>(p, vec, s);
        case VMS_F16: return dispatch_cfwd<f16_t>(p, vec, s);
        default: return dispatch_cfwd<bf16_t>(p, vec, s);
    }
}

int launch_scan_bwd_complex(const vms_scan_bwd_params& q, bool vec, hipStream_t s) {
    const vms_scan_fwd_params& p = q.f;
    if (int rc = validate_complex(p)) return rc;
    VMS_CHECK(p.seqlen <= kCCS || (p.x != nullptr && p.x_has_sub == 1),
              "complex A: the backward needs the forward's 512-element checkpoints (x with x_has_sub == 1) when seqlen > 512");
    set_last_kernel("scan_bwd_complex");
    switch (p.dtype) {
        case VMS_F32: return dispatch_cbwd<float>(q, vec, s);
        case VMS_F16: return dispatch_cbwd<f16_t>(q, vec, s);
        default: return dispatch_cbwd<bf16_t>(q, vec, s);
    }
}

}  // namespace vms
