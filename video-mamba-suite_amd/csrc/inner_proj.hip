// inner_proj.hip -- the SMALL projections of the Mamba inner node on the matrix cores of gfx950 (wave64, MFMA 32x32x16).
//
// Between conv1d and the scan the reference runs four skinny GEMMs per direction on (batch, channels, seqlen) activations
// (mamba_ssm/ops/selective_scan_interface.py): forward  delta = dt_proj.weight @ x_dbl[:R]               (:182),
// backward  ddt_proj.weight = ddelta @ x_dbl[:R]^T (:275),  dx_dbl[:R] = dt_proj.weight^T @ ddelta        (:276),
//           dx_proj.weight = dx_dbl @ conv1d_out^T  (:278),  dconv1d_out += x_proj.weight^T @ dx_dbl      (:279).
// One side of each is 24..96 wide, the other is the 134 MB activation: they are HBM streaming problems with a little
// matrix work attached, and the library's general GEMM kernels run them at 1.3-2x their memory floor
// (profiles/r03_small_gemms.md).  Two hand-written kernels cover them:
//
//   proj_apply   out[b][d][l] (+)= sum_r W[d][r] in[b][r][l]        K = r <= 96 all on chip; out / in with unit l stride.
//                A workgroup = 128 rows d x 64 positions l per step, walking a range of l; the `in` tile (K x 64, shared by
//                the 4 waves) goes global -> registers -> LDS row-major and becomes the MFMA B operand through
//                ds_read_b64_tr_b16 (K is the STRIDED axis of `in`: the transposing LDS read is what makes the (b, r, l)
//                layout usable as it is); W fragments live in registers for the whole range; the 32 x 64 fp32 result of a
//                wave is turned from the MFMA C layout (lane = column) into row pieces through a wave-private LDS tile so
//                that the optional read-modify-write of `out` and its store are 16-byte row-contiguous accesses.
//   proj_wgrad   dW[m][n] += sum_{b, l} P[b][m][l] Q[b][n][l]        K = l (both operands K-contiguous); m <= 128.
//                A workgroup = 128 rows n x a range of l of one batch entry; P (shared) and Q (a wave's 32 rows) tiles of
//                64 positions go through LDS in full 128-byte lines (fragment-shaped global loads would fetch every line
//                8 times through the L1); fp32 accumulators for all of m stay in registers over the range; one fp32
//                atomic per (m, n) and workgroup at the end (dW is the caller's zero-filled fp32 buffer, like dA / dD).
#include "vms_common.h"

namespace vms {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) s16x8 lds_s16x8;
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;

template <typename T> struct Mfma32;
template <> struct Mfma32<bf16_t> {
    typedef __bf16 V __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ f32x16 run(s16x8 a, s16x8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(V, a), __builtin_bit_cast(V, b), c, 0, 0, 0);
    }
};
template <> struct Mfma32<f16_t> {
    typedef _Float16 V __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ f32x16 run(s16x8 a, s16x8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(V, a), __builtin_bit_cast(V, b), c, 0, 0, 0);
    }
};

#ifndef VMS_PCB_ST_AUX
#define VMS_PCB_ST_AUX 0   /* the same for proj_conv_bwd's dx stores */
#endif
#ifndef VMS_CXP_ST_AUX
#define VMS_CXP_ST_AUX 0   /* cache policy bits of the conv1d_out stores (A/B builds: 1 = sc0, 2 = nt / streaming, 3) */
#endif
constexpr int kPBufFlags = 0x00020000;   // gfx9 raw buffer, 32-bit data format (as causal_conv1d.hip)
constexpr int kPT = 256;           // threads per workgroup (4 waves)
constexpr int kTL = 64;            // positions per tile
constexpr int kRowE = kTL + 8;     // LDS row pitch of a 16-bit tile, elements (144 B: 16 consecutive rows = 16 distinct 16-byte slots)
constexpr int kEpE = 32 + 4;       // LDS row pitch of the fp32 epilogue half tile, floats
// Both kernels are latency-bound streams (a wave has one tile of loads in flight), so what counts is resident waves: single
// LDS buffers (two workgroup barriers per tile instead of one) and <= 128 registers give 4 workgroups = 16 waves per CU.
#define VMS_PROJ_BOUNDS __launch_bounds__(kPT, 4)

__device__ __forceinline__ bool aligned16_dev(const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15) == 0; }
__device__ __forceinline__ void lds_order() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// ---------------------------------------------------------------------------------------------------------------------
// proj_apply
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int KS, bool ACC>
__global__ VMS_PROJ_BOUNDS void proj_apply_kernel(const vms_proj_apply_params p, const int tiles_per_wg) {
    constexpr int KR = KS * 16;
    constexpr int NPASS = (KR + 31) / 32;   // `in` tile: 32 rows of 8 x 16-byte pieces per pass of the workgroup
    typedef __attribute__((address_space(3))) short lds_s16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, c32 = lane & 31;
    const int b = blockIdx.z, d0 = blockIdx.y * 128 + wave * 32;
    const int L = p.seqlen, R = p.k;
    const T* const in_b = static_cast<const T*>(p.in) + (int64_t)b * p.in_batch_stride;
    T* const out_b = static_cast<T*>(p.out) + (int64_t)b * p.out_batch_stride;
    lds_s16* const in_lds = (lds_s16*)reinterpret_cast<short*>(smem);                                                    // [KR][kRowE]
    lds_f32* const ep = (lds_f32*)reinterpret_cast<float*>(smem + KR * kRowE * 2) + wave * (32 * kEpE);                  // [32][kEpE], this wave's

    // W fragments: A[i = c32][k = 16 s + 8 h + e] = W[d0 + c32][k], zero beyond the matrix
    s16x8 wf[KS];
    // (dword-aligned rows: the range check of a buffer load works on dwords -- a row ending inside one would lose its last element)
    if (p.w_k_stride == 1 && R % 2 == 0 && p.w_row_stride % 2 == 0 && (reinterpret_cast<uintptr_t>(p.w) & 3) == 0 &&
        (int64_t)p.rows * p.w_row_stride * 2 < ((int64_t)1 << 31)) {
        // k contiguous (dt_proj.weight): a fragment row piece is ONE 16-byte buffer load (any 2-byte alignment), all KS of them in
        // flight together -- gathered element by element the 8 KS loads of a lane came out as KS x 4 dependent round trips at
        // the start of every workgroup (a quarter of its life at 5 tiles per workgroup)
        const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<void*>(p.w), 0, (int)(((int64_t)(p.rows - 1) * p.w_row_stride + R) * 2), 0x00020000);
        const int d = d0 + c32;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int k = 16 * s + 8 * h;
            const s16x8 v = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(w_rs, d < p.rows && k < R ? (int)(((int64_t)d * p.w_row_stride + k) * 2) : -1, 0, 0));
#pragma unroll
            for (int e = 0; e < 8; ++e) wf[s][e] = k + e < R ? v[e] : (short)0;     // (what follows the row's end is the next row)
        }
    } else {
        const int d = d0 + c32;
        const T* wrow = static_cast<const T*>(p.w) + (int64_t)(d < p.rows ? d : 0) * p.w_row_stride;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = 16 * s + 8 * h + e;
                const T v = (d < p.rows && k < R) ? wrow[(int64_t)k * p.w_k_stride] : static_cast<T>(0.f);
                wf[s][e] = __builtin_bit_cast(short, v);
            }
    }
    const int t_lo = blockIdx.x * tiles_per_wg;
    const int n_tiles = (L + kTL - 1) / kTL;
    const int t_hi = t_lo + tiles_per_wg < n_tiles ? t_lo + tiles_per_wg : n_tiles;
    if (t_lo >= t_hi) return;

    // staging of the `in` tile: thread -> row (tid >> 3) + 32 pass, piece tid & 7.  Every access of the loop goes through a buffer
    // resource per tensor and batch entry (out-of-range offset: loads return zeros, stores are dropped) instead of a select or a
    // branch: the compiler then counts what is in flight -- with plain loads carried over the loop's back edge it drained everything,
    // the tile's STORES included, at every iteration (vmcnt(0): 66 % of the wave cycles waiting) -- and the `in` tiles of two steps
    // ahead travel while a tile computes.
    const __amdgpu_buffer_rsrc_t in_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(in_b), 0, (int)(((int64_t)(R - 1) * p.in_k_stride + L) * 2), kPBufFlags);
    const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(out_b, 0, (int)(((int64_t)(p.rows - 1) * p.out_row_stride + L) * 2), kPBufFlags);
    typedef unsigned int au32x4 __attribute__((ext_vector_type(4)));
    struct Stage { s16x8 v[NPASS]; };
    auto stage_load = [&](Stage& st, int t) __attribute__((always_inline)) {
        const int l = t * kTL + 8 * (tid & 7);
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int r = (tid >> 3) + 32 * ps;
            const bool ok = r < R && l < L && t < t_hi;
            st.v[ps] = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(in_rs, ok ? (int)(((int64_t)r * p.in_k_stride + l) * 2) : -1, 0, 0));
        }
    };
    auto stage_store = [&](const Stage& st) __attribute__((always_inline)) {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int r = (tid >> 3) + 32 * ps;
            if (NPASS * 32 == KR || r < KR) *reinterpret_cast<lds_s16x8*>(in_lds + r * kRowE + 8 * (tid & 7)) = st.v[ps];
        }
    };
    // transposing read: lane i of a 16-lane group supplies the address of row i / 4, columns 4 (i % 4) .. + 3 of a [4][16] block
    // and receives column i of its 4 rows (tools/microbench/tr_probe.hip).  Groups 0 / 1 = columns 0-15 / 16-31 of k rows 8 h .. 8 h + 3 (+ 4)
    const int i16 = lane & 15, g16 = lane >> 4;
    const lds_s16* const tb = in_lds + (8 * (g16 >> 1) + (i16 >> 2)) * kRowE + 16 * (g16 & 1) + 4 * (i16 & 3);
    // epilogue pieces of a 32 x 32 half tile: lane -> row (lane >> 2) + 16 pp, columns 8 (lane & 3) .. + 7
    const int er = lane >> 2, ec = 8 * (lane & 3);

    auto step = [&](Stage& st, int t) __attribute__((always_inline)) {
        const int l0 = t * kTL;
        stage_store(st);
        __syncthreads();          // tile t is in LDS
        stage_load(st, t + 2);    // travels during this tile and the next
        vec_t<T, 8> prev[2][2];
        if (ACC) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    const int d = d0 + er + 16 * pp, l = l0 + 32 * j + ec;
                    prev[j][pp] = __builtin_bit_cast(vec_t<T, 8>, __builtin_amdgcn_raw_buffer_load_b128(
                        out_rs, d < p.rows && l < L ? (int)(((int64_t)d * p.out_row_stride + l) * 2) : -1, 0, 0));
                }
        }
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tb + (16 * s) * kRowE + 32 * j));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tb + (16 * s + 4) * kRowE + 32 * j));
                const s16x8 bf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                acc[j] = Mfma32<T>::run(wf[s], bf, acc[j]);
            }
        }
        __syncthreads();          // every wave has read tile t: the buffer may be overwritten
        // C layout (column = c32, row = (v & 3) + 8 (v >> 2) + 4 h) -> this wave's fp32 half tile -> row pieces
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            lds_order();   // the previous half's reads of `ep` are done
#pragma unroll
            for (int v = 0; v < 16; ++v) ep[((v & 3) + 8 * (v >> 2) + 4 * h) * kEpE + c32] = acc[j][v];
            lds_order();
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const int row = er + 16 * pp, d = d0 + row, l = l0 + 32 * j + ec;
                const lds_f32x4* src = (const lds_f32x4*)(ep + row * kEpE + ec);
                const f32x4 a = src[0], c = src[1];
                float f[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
                vec_t<T, 8> o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (ACC) f[e] += static_cast<float>(prev[j][pp][e]);
                    o[e] = static_cast<T>(f[e]);
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(au32x4, o), out_rs,
                                                       d < p.rows && l < L ? (int)(((int64_t)d * p.out_row_stride + l) * 2) : -1, 0, 0);
            }
        }
    };
    Stage sa, sb;
    stage_load(sa, t_lo);
    stage_load(sb, t_lo + 1);
    for (int t = t_lo; t < t_hi; t += 2) {
        step(sa, t);
        if (t + 1 < t_hi) step(sb, t + 1);     // (workgroup-uniform)
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// proj_wgrad
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int MB>
__global__ __launch_bounds__(kPT, (MB <= 2 ? 4 : 2)) void proj_wgrad_kernel(const vms_proj_wgrad_params p, const int tiles_per_wg) {
    constexpr int MR = MB * 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) short lds_s16;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, c32 = lane & 31;
    const int b = blockIdx.z, n0 = blockIdx.y * 128 + wave * 32;
    const int L = p.seqlen;
    const T* const P_b = static_cast<const T*>(p.p) + (int64_t)b * p.p_batch_stride;
    const T* const Q_b = static_cast<const T*>(p.q) + (int64_t)b * p.q_batch_stride;
    lds_s16* const p_lds = (lds_s16*)reinterpret_cast<short*>(smem);                                   // [MR][kRowE]
    lds_s16* const q_lds = p_lds + MR * kRowE + wave * (32 * kRowE);                                   // [32][kRowE], this wave's

    const int n_tiles = (L + kTL - 1) / kTL;
    const int t_lo = blockIdx.x * tiles_per_wg;
    const int t_hi = t_lo + tiles_per_wg < n_tiles ? t_lo + tiles_per_wg : n_tiles;
    if (t_lo >= t_hi) return;

    // Every load of the loop goes through a buffer resource (out-of-range offset = zeros, no memory access) instead of a select:
    // the compiler counts the requests, so the tiles of TWO steps ahead stay in flight across the barriers (the grid is 1-2 waves per
    // SIMD on the d = 768 shapes: with one tile of requests per wave the kernel ran at 2.3 TB/s)
    const __amdgpu_buffer_rsrc_t p_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(P_b), 0, (int)(((int64_t)(p.m - 1) * p.p_row_stride + L) * 2), kPBufFlags);
    const __amdgpu_buffer_rsrc_t q_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Q_b), 0, (int)(((int64_t)(p.n - 1) * p.q_row_stride + L) * 2), kPBufFlags);
    struct Stage { s16x8 p[MB], q[4]; };
    auto stage_load = [&](Stage& st, int t) __attribute__((always_inline)) {
        const int lp = t * kTL + 8 * (tid & 7), lq = t * kTL + 8 * (lane & 7);
#pragma unroll
        for (int ps = 0; ps < MB; ++ps) {
            const int r = (tid >> 3) + 32 * ps;
            const bool ok = r < p.m && lp < L && t < t_hi;
            st.p[ps] = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(p_rs, ok ? (int)(((int64_t)r * p.p_row_stride + lp) * 2) : -1, 0, 0));
        }
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int n = n0 + (lane >> 3) + 8 * ps;
            const bool ok = n < p.n && lq < L && t < t_hi;
            st.q[ps] = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(q_rs, ok ? (int)(((int64_t)n * p.q_row_stride + lq) * 2) : -1, 0, 0));
        }
    };
    auto stage_store = [&](const Stage& st) __attribute__((always_inline)) {
#pragma unroll
        for (int ps = 0; ps < MB; ++ps)
            *reinterpret_cast<lds_s16x8*>(p_lds + ((tid >> 3) + 32 * ps) * kRowE + 8 * (tid & 7)) = st.p[ps];
#pragma unroll
        for (int ps = 0; ps < 4; ++ps)
            *reinterpret_cast<lds_s16x8*>(q_lds + ((lane >> 3) + 8 * ps) * kRowE + 8 * (lane & 7)) = st.q[ps];
    };
    f32x16 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[mb][v] = 0.f;

    const lds_s16* const pa = p_lds + c32 * kRowE + 8 * h;
    const lds_s16* const qa = q_lds + c32 * kRowE + 8 * h;
    auto product = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < kTL / 16; ++s) {
            const s16x8 bq = *reinterpret_cast<const lds_s16x8*>(qa + 16 * s);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const s16x8 ap = *reinterpret_cast<const lds_s16x8*>(pa + 32 * mb * kRowE + 16 * s);
                acc[mb] = Mfma32<T>::run(ap, bq, acc[mb]);
            }
        }
    };
    Stage sa, sb;
    stage_load(sa, t_lo);
    stage_load(sb, t_lo + 1);
    for (int t = t_lo; t < t_hi; t += 2) {
        stage_store(sa);
        __syncthreads();          // tile t is in LDS
        stage_load(sa, t + 2);
        product();
        __syncthreads();          // every wave has read tile t
        if (t + 1 < t_hi) {       // (workgroup-uniform; no load under it)
            stage_store(sb);
            __syncthreads();
        }
        stage_load(sb, t + 3);
        if (t + 1 < t_hi) {
            product();
            __syncthreads();
        }
    }
    if (!p.dw_transposed) {
        const int n = n0 + c32;
        if (n < p.n) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int m = 32 * mb + (v & 3) + 8 * (v >> 2) + 4 * h;
                    if (m < p.m) atomicAdd(p.dw + (int64_t)m * p.dw_row_stride + n, acc[mb][v]);
                }
        }
        return;
    }
    // dw stored (n, m) -- dt_proj.weight's own (d_inner, dt_rank) layout, so that autograd keeps the gradient instead of copying a
    // transposed view: 32 rows m at a time through an LDS tile [n][m], atomics with the lanes along m (128 contiguous bytes per half wave)
    lds_f32* const tt = (lds_f32*)reinterpret_cast<float*>(smem);   // [128][33] floats: the staging tiles are free (last barrier of the loop)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
        for (int v = 0; v < 16; ++v) tt[(wave * 32 + c32) * 33 + (v & 3) + 8 * (v >> 2) + 4 * h] = acc[mb][v];
        __syncthreads();
        const int ml = tid & 31, m = 32 * mb + ml;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int nl = (tid >> 5) + 8 * i, n = blockIdx.y * 128 + nl;
            if (m < p.m && n < p.n) atomicAdd(p.dw + (int64_t)n * p.dw_row_stride + m, tt[nl * 33 + ml]);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// proj_conv_bwd: the tail of the inner node's backward in one pass
// ---------------------------------------------------------------------------------------------------------------------
// After the scan's backward the reference runs  dx_proj_weight = dx_dbl @ conv1d_out^T (SSI:278),  dconv1d_out += x_proj_weight^T
// @ dx_dbl (:279)  and causal_conv1d_bwd (:283) -- three kernels that read or write the (batch, dim, seqlen) activations
// seven times (conv1d_out once, dconv1d_out read + written + read, x once, dx written, + the recompute of conv1d_out's
// pre-activation inside the conv backward).  Here a workgroup (4 waves x 16 channels) walks 64 channels x 64 positions at a
// time and does all three on chip:
//   * T = W_x^T dx_dbl on the matrix cores (16x16x32; both operands k-strided in memory: transposing LDS reads of the dx_dbl
//     tile and of the W_x tile, which stays in LDS for the workgroup's whole range), through a wave-private fp32 LDS tile into
//     8-position row pieces: lane = (channel row lane >> 3, piece lane & 7), so a wave's load / store covers 8 full 128-byte lines;
//   * g = T + du and the conv backward on the pieces: pre-activation, SiLU', dx, dweight / dbias exactly as causal_conv1d_bwd
//     computes them; dconv1d_out is never materialised (fp32 in registers: one rounding less than the reference);
//   * conv1d_out = silu(pre) rounded to the activation dtype into a wave-private LDS tile = the B operand of the second
//     product, dW_x += dx_dbl conv1d_out^T (16x16x32, fp32 accumulators over the range, one atomic per (k, channel) at the end).
// HBM traffic: du and x read, dx written, dx_dbl from L2 -- the conv backward's own three passes.
// Tiles are walked from the end of the (logical) sequence: dx needs SiLU' * g of the next 3 positions, which the previous
// iteration left in the piece-0 lanes (`carry`); the first iteration of a workgroup's range recomputes them from the tile after
// its range without storing anything.  Right-to-left rows (reverse / reverse_from) use the same code on mirrored addresses:
// a piece is kept in PHYSICAL element order everywhere (LDS tiles, MFMA columns) and read through (REV ? 7 - i : i) where the
// conv needs logical order.
// Where the time goes (profiles/r03_small_gemms.md, r03_sq_tail.md): 2 waves per SIMD (176-188 VGPRs: the dW_x accumulators, a
// tile of requests in flight, the conv's working set), the VALU pipe 72 % busy with 36 instructions per element (the stand-alone
// conv backward: 34), HBM traffic 1.20x the algorithmic bytes: bound by vector issue, 114-119 us at (8, 1024, 8192) where its
// three passes would take 75 at the chip's streaming rate.  A 3-waves-per-SIMD build spills and is slower.
typedef unsigned int pu32x4 __attribute__((ext_vector_type(4)));
constexpr int kCD = 64;             // channels per workgroup of proj_conv_bwd (16 per wave)
constexpr int kWRowE = kCD + 8;     // LDS row pitch of the W_x tile (k rows x 64 channels), elements
constexpr int kEpT = kTL + 4;       // LDS row pitch of the fp32 tile of the first product (16 channels x 64 positions), floats

template <typename T> struct Mfma16;
template <> struct Mfma16<bf16_t> {
    typedef __bf16 V __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ f32x4 run(s16x8 a, s16x8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(V, a), __builtin_bit_cast(V, b), c, 0, 0, 0);
    }
};
template <> struct Mfma16<f16_t> {
    typedef _Float16 V __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ f32x4 run(s16x8 a, s16x8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(V, a), __builtin_bit_cast(V, b), c, 0, 0, 0);
    }
};

template <typename T, int KS, bool REV, bool DXACC, bool RAG>
__device__ __forceinline__ void proj_conv_bwd_body(const vms_proj_conv_bwd_params& p, const int tiles_per_wg, const int2 p_grid, const int entry) {
    constexpr int MB = (KS * 16 + 31) / 32;      // 32-deep k steps of the first product
    constexpr int KP = MB * 32;                  // k padded to the matrix instructions' depth (rows beyond k are zero)
    constexpr int NPASS = MB;                    // dx_dbl tile: 32 rows of 8 x 16-byte pieces per pass of the workgroup
    typedef __attribute__((address_space(3))) short lds_s16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 1-D grid, decoded so that the workgroups that share a dx_dbl range -- the channel tiles of one (batch entry, range of
    // positions) -- run on ONE XCD (workgroup id % 8 on this chip) and re-read it from that XCD's L2: with a (ranges, channel
    // tiles, batch) grid they were spread over all 8 and every L2 fetched its own copy (FETCH_SIZE 394 MB for 281 MB of inputs)
    // (consecutive ids of one XCD = the channel tiles of one range: they are resident together)
    const int n_rng = p_grid.x, n_dt = p_grid.y, n_pairs = n_rng * p.batch;
    const int slot = (int)blockIdx.x >> 3, pair = (slot / n_dt) * 8 + ((int)blockIdx.x & 7);
    if (pair >= n_pairs) return;
    const int bid_x = pair % n_rng, b = pair / n_rng, bid_y = slot % n_dt;
    const int d0w = bid_y * kCD, d0 = d0w + wave * 16;
    const int L = p.seqlen, R = p.k;
    const T* const in_b = static_cast<const T*>(p.dx_dbl) + (int64_t)b * p.dxdbl_batch_stride;
    lds_s16* const in_lds = (lds_s16*)reinterpret_cast<short*>(smem);                                                      // [KP][kRowE]
    lds_f32* const ep = (lds_f32*)reinterpret_cast<float*>(smem + KP * kRowE * 2) + wave * (16 * kEpT);                    // [16][kEpT] fp32, this wave's
    lds_s16* const co = (lds_s16*)reinterpret_cast<short*>(smem + KP * kRowE * 2 + 4 * 16 * kEpT * 4) + wave * (16 * kRowE);   // [16][kRowE], this wave's
    lds_s16* const w_lds = (lds_s16*)reinterpret_cast<short*>(smem + KP * kRowE * 2 + 4 * 16 * kEpT * 4 + 4 * 16 * kRowE * 2);   // [KP][kWRowE]

    const int n_tiles = (L + kTL - 1) / kTL;
    const int t_lo = bid_x * tiles_per_wg;
    const int t_hi = t_lo + tiles_per_wg < n_tiles ? t_lo + tiles_per_wg : n_tiles;
    if (t_lo >= t_hi) return;

    // W_x[:, d0w .. d0w + 64) as it is stored (row k, channels contiguous), zero beyond the matrix; the A operand of the first
    // product (i = channel, k strided) comes out of it through the transposing read, like the B operand out of the dx_dbl tile
    {
        const T* const wx = static_cast<const T*>(p.w_x);
        const bool vec = p.wx_c_stride == 1 && p.wx_k_stride % 8 == 0 && aligned16_dev(wx) && d0w + kCD <= p.dim;
        if (vec) {   // 8 lanes x 16 bytes per row: KP / 32 independent loads per thread
#pragma unroll
            for (int it = 0; it < KP / 32; ++it) {
                const int k = it * 32 + (tid >> 3), c = 8 * (tid & 7);
                const bool okw = k < R;
                const s16x8 v = *reinterpret_cast<const s16x8*>(wx + (int64_t)(okw ? k : 0) * p.wx_k_stride + d0w + c);
                *reinterpret_cast<lds_s16x8*>(w_lds + k * kWRowE + c) = okw ? v : s16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
        } else {
            for (int idx = tid; idx < KP * kCD; idx += kPT) {
                const int k = idx / kCD, c = idx % kCD;
                const bool okw = k < R && d0w + c < p.dim;
                const T v = okw ? wx[(int64_t)k * p.wx_k_stride + (int64_t)(d0w + c) * p.wx_c_stride] : static_cast<T>(0.f);
                w_lds[k * kWRowE + c] = __builtin_bit_cast(short, v);
            }
        }
    }

    // A lane's pieces: 8 lanes x 16 bytes cover the 64 positions of one channel row of the tile (full 128-byte lines: with 4
    // lanes per row every load fetched 16 half lines and the kernel streamed at 2.8 TB/s with all of its arithmetic removed);
    // sub-step ss = 0, 1 handles channel row 8 ss + rr of the wave's 16.
    const int rr = lane >> 3, pc = lane & 7, ec = 8 * pc;
    // Every global access of the loop goes through a buffer resource per tensor and batch entry with an out-of-range offset
    // for what must not be touched (rows beyond dim, positions beyond seqlen, tiles beyond the range): such loads return 0 and
    // such stores are dropped, so NO access sits under a branch or feeds a select -- the compiler can count them and waits
    // with s_waitcnt vmcnt(N) for exactly the loads it needs (with predicated accesses it drained everything, vmcnt(0), at
    // every barrier: the "tile ahead" requests then had a quarter of an iteration to land and the kernel took 147 us).
    const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(static_cast<const T*>(p.x) + (int64_t)b * p.x_batch_stride), 0, (int)(p.dim * p.x_c_stride * 2), kPBufFlags);
    const __amdgpu_buffer_rsrc_t du_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(static_cast<const T*>(p.du) + (int64_t)b * p.du_batch_stride), 0, (int)(p.dim * p.du_c_stride * 2), kPBufFlags);
    const __amdgpu_buffer_rsrc_t dx_rs = __builtin_amdgcn_make_buffer_rsrc(
        static_cast<T*>(p.dx) + (int64_t)b * p.dx_batch_stride, 0, (int)(p.dim * p.dx_c_stride * 2), kPBufFlags);
    const __amdgpu_buffer_rsrc_t in_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(in_b), 0, (int)(R * p.dxdbl_k_stride * 2), kPBufFlags);
    constexpr int kOOB = -1;    // as an unsigned byte offset: beyond every buffer
    float taps[2][4], cbias[2];
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
        const int d = d0 + 8 * ss + rr;
        const int dc = d < p.dim ? d : 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {   // taps[k] multiplies x[t - 3 + k]; widths < 4 get leading zeros (causal_conv1d.hip load_taps)
            const int w = k - (4 - p.width);
            float v = 0.f;
            if (w >= 0) {
                const int64_t idx = (int64_t)dc * p.conv_weight_c_stride + (int64_t)w * p.conv_weight_width_stride;
                v = p.wdtype == VMS_F32 ? static_cast<const float*>(p.conv_weight)[idx]
                    : p.wdtype == VMS_F16 ? static_cast<float>(static_cast<const f16_t*>(p.conv_weight)[idx])
                                          : static_cast<float>(static_cast<const bf16_t*>(p.conv_weight)[idx]);
            }
            taps[ss][k] = v;
        }
        cbias[ss] = !p.conv_bias ? 0.f
                    : p.wdtype == VMS_F32 ? static_cast<const float*>(p.conv_bias)[dc]
                    : p.wdtype == VMS_F16 ? static_cast<float>(static_cast<const f16_t*>(p.conv_bias)[dc])
                                          : static_cast<float>(static_cast<const bf16_t*>(p.conv_bias)[dc]);
    }

    // A piece = the 8 logical positions [tl, tl + 8) of a row, kept in PHYSICAL element order.  RAG (seqlen % 8 != 0, or rows that
    // are not 16-byte aligned): gfx950 serves 16-byte buffer accesses at any 2-byte alignment, so whole pieces move as before;
    // the row's single partly valid piece (nv = seqlen - tl < 8 positions) moves element by element and reads as zeros beyond the
    // row, which is all the arithmetic needs (as for rows / tiles out of range).  `row` = element offset of the row in the buffer.
    auto ld8 = [&](const __amdgpu_buffer_rsrc_t& rs, int64_t row, int tl, bool ok) __attribute__((always_inline)) -> s16x8 {
        const int pl = REV ? L - tl - 8 : tl;
        if (!RAG || L - tl >= 8 || !ok)
            return __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? (int)((row + pl) * 2) : kOOB, 0, 0));
        s16x8 v;
#pragma unroll
        for (int i = 0; i < 8; ++i) {     // logical element i = physical element (REV ? 7 - i : i) of the piece
            const int phys = REV ? L - 1 - (tl + i) : tl + i;
            v[REV ? 7 - i : i] = (short)__builtin_amdgcn_raw_buffer_load_b16(rs, tl + i < L ? (int)((row + phys) * 2) : kOOB, 0, 0);
        }
        return v;
    };
    auto st8 = [&](const __amdgpu_buffer_rsrc_t& rs, int64_t row, int tl, bool ok, const s16x8& v) __attribute__((always_inline)) {
        const int pl = REV ? L - tl - 8 : tl;
        if (!RAG || L - tl >= 8 || !ok) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pu32x4, v), rs, ok ? (int)((row + pl) * 2) : kOOB, 0, VMS_PCB_ST_AUX);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int phys = REV ? L - 1 - (tl + i) : tl + i;
                __builtin_amdgcn_raw_buffer_store_b16((unsigned short)v[REV ? 7 - i : i], rs, tl + i < L ? (int)((row + phys) * 2) : kOOB, 0, 0);
            }
        }
    };

    // staging of the dx_dbl tile (logical tile t): thread -> row (tid >> 3) + 32 pass, logical piece tid & 7
    s16x8 stg[NPASS];
    auto stage_load = [&](int t) __attribute__((always_inline)) {
        const int tl = t * kTL + 8 * (tid & 7);                      // logical start of the piece
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int r = (tid >> 3) + 32 * ps;
            stg[ps] = ld8(in_rs, (int64_t)r * p.dxdbl_k_stride, tl, r < R && tl < L && t >= t_lo);
        }
    };
    auto stage_store = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps)
            *reinterpret_cast<lds_s16x8*>(in_lds + ((tid >> 3) + 32 * ps) * kRowE + 8 * (tid & 7)) = stg[ps];
    };
    // transposing reads (tools/microbench/tr_probe.hip): lane i of a 16-lane group supplies row i / 4, columns 4 (i % 4) .. + 3 of a
    // [4][16] block and receives column i of its 4 rows.  16x16x32 operands: lane -> (row / column lane & 15, k = 8 (lane >> 4) .. + 7)
    const int i16 = lane & 15, g16 = lane >> 4;
    const lds_s16* const tb = in_lds + (8 * g16 + (i16 >> 2)) * kRowE + 4 * (i16 & 3);
    const lds_s16* const ta = w_lds + (8 * g16 + (i16 >> 2)) * kWRowE + wave * 16 + 4 * (i16 & 3);

    f32x4 accw[KP / 16];    // dW_x[16 mb + 4 g16 + v][d0 + i16]
#pragma unroll
    for (int mb = 0; mb < KP / 16; ++mb) accw[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dwacc[2][4], dbacc[2], carry[2][3];   // carry: SiLU' g of the first 3 positions after this row's last piece (piece-0 lanes)
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
        dbacc[ss] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) dwacc[ss][k] = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) carry[ss][k] = 0.f;
    }

    // The x / du vectors of a tile's two pieces are requested a whole tile ahead, at the top of the previous iteration, and
    // WIDENED at its bottom: nothing loaded is carried across the loop's back edge (hipcc waits for loop-carried loads with a
    // full s_waitcnt vmcnt(0) at the loop head, which also drains the dx stores of the iteration), so the waits are counted
    // and leave the stores in flight.
    struct Raw { vec_t<T, 8> x, du; vec_t<T, 4> xh; };
    Raw raw[2];
    float xvc[2][8 + 3], duc[2][8];     // inputs t - 3 .. t + 7 and du of the piece, logical order, of the tile being processed
    auto widen = [&](int ss) __attribute__((always_inline)) {
        // logical order: element i of the piece is physical element (REV ? 7 - i : i)
#pragma unroll
        for (int k = 0; k < 3; ++k) xvc[ss][k] = static_cast<float>(raw[ss].xh[REV ? 2 - k : 1 + k]);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            xvc[ss][3 + i] = static_cast<float>(raw[ss].x[REV ? 7 - i : i]);
            duc[ss][i] = static_cast<float>(raw[ss].du[REV ? 7 - i : i]);
        }
    };
    auto piece_load = [&](Raw& r, int t, int ss) __attribute__((always_inline)) {
        const int d = d0 + 8 * ss + rr;
        const int tl = t * kTL + ec;
        const int pl = REV ? L - tl - 8 : tl;
        const bool okp = d < p.dim && tl < L && t >= t_lo;
        const int xo = (int)((d * p.x_c_stride + pl) * 2);
        r.x = __builtin_bit_cast(vec_t<T, 8>, ld8(x_rs, (int64_t)d * p.x_c_stride, tl, okp));
        r.du = __builtin_bit_cast(vec_t<T, 8>, ld8(du_rs, (int64_t)d * p.du_c_stride, tl, okp));
        // the 3 positions before the piece: physical [pl - 4, pl) left-to-right, [pl + 8, pl + 12) right-to-left (always inside the
        // row: tl >= 8 there); zeros before the row
        // entry (> 0: a power of two, a multiple of 8): the row is `seqlen / entry` independent sequences of `entry` positions laid end
        // to end (vms_proj_conv_bwd folds gap-free batches of short sequences into one row): nothing crosses their boundaries
        r.xh = __builtin_bit_cast(vec_t<T, 4>, __builtin_amdgcn_raw_buffer_load_b64(x_rs, okp && tl > 0 && (entry == 0 || (tl & (entry - 1)) != 0) ? xo + (REV ? 16 : -8) : kOOB, 0, 0));
    };

    // the tile after the range only feeds `carry`
    const int t_first = t_hi < n_tiles ? t_hi : t_hi - 1;
    stage_load(t_first);
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) piece_load(raw[ss], t_first, ss);
    stage_store();
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) widen(ss);
    for (int t = t_first; t >= t_lo; --t) {
        const bool emit = t < t_hi;
        __syncthreads();          // tile t (and, the first time, the W_x tile) is in LDS
        stage_load(t - 1);        // travel during the rest of the iteration
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) piece_load(raw[ss], t - 1, ss);
        const int tl = t * kTL + ec;
        vec_t<T, 8> dxold[2];
        if (DXACC) {
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                const int d = d0 + 8 * ss + rr;
                const bool okp = d < p.dim && tl < L && emit;
                dxold[ss] = __builtin_bit_cast(vec_t<T, 8>, ld8(dx_rs, (int64_t)d * p.dx_c_stride, tl, okp));
            }
        }
        {
            // first product: this wave's 16 channels x the tile's 64 positions, into the wave's fp32 tile
            f32x4 acc[4];
#pragma unroll
            for (int s = 0; s < MB; ++s) {
                const s16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ta + (32 * s) * kWRowE));
                const s16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ta + (32 * s + 4) * kWRowE));
                const s16x8 af = __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tb + (32 * s) * kRowE + 16 * nb));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tb + (32 * s + 4) * kRowE + 16 * nb));
                    const s16x8 bf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    acc[nb] = Mfma16<T>::run(af, bf, s == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[nb]);
                }
            }
            lds_order();   // the previous tile's reads of `ep` are done
            // C layout: column = lane & 15, row = 4 (lane >> 4) + register
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int v = 0; v < 4; ++v) ep[(4 * g16 + v) * kEpT + 16 * nb + i16] = acc[nb][v];
            lds_order();
        }
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            const int d = d0 + 8 * ss + rr;
            const bool okp = d < p.dim && tl < L;
            float gp[8 + 3];                 // SiLU' g of t .. t + 10, logical order
            const float (&xv)[8 + 3] = xvc[ss];
            const float (&duf)[8] = duc[ss];
            const lds_f32x4* src = (const lds_f32x4*)(ep + (8 * ss + rr) * kEpT + ec);
            const f32x4 tq0 = src[0], tq1 = src[1];
            const float tv[8] = {tq0.x, tq0.y, tq0.z, tq0.w, tq1.x, tq1.y, tq1.z, tq1.w};
            vec_t<T, 8> cov;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = REV ? 7 - i : i;
                float pre = cbias[ss];
#pragma unroll
                for (int k = 0; k < 4; ++k) pre = fmaf(taps[ss][k], xv[i + k], pre);
                const float sg = sigmoidf_(pre);
                // beyond the row / the sequence every load returned 0: T = 0 (zero dx_dbl columns, zero W_x rows), du = 0, so
                // g = 0 and gp = 0 without a select; conv1d_out is silu(bias) there but meets a zero dx_dbl column (or a dW_x
                // column that is never written) in the second product
                const float g = tv[e] + duf[i];
                gp[i] = g * (sg * (1.f + pre * (1.f - sg)));
                cov[e] = static_cast<T>(pre * sg);
            }
            *reinterpret_cast<__attribute__((address_space(3))) vec_t<T, 8>*>(co + (8 * ss + rr) * kRowE + ec) = cov;
            // the next piece's first three values: lane + 1 (the 8 lanes of a channel row are contiguous); the row's last piece
            // takes what piece 0 of the previously processed tile left in `carry` (7 lanes down)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float fromnext = dpp_mov<DPP_ROW_SHL1, 0xf>(0.f, gp[k]);
                const float fromcarry = dpp_mov<0x117, 0xf>(0.f, carry[ss][k]);   // row_shr:7
                gp[8 + k] = entry != 0 && ((tl + 8) & (entry - 1)) == 0 ? 0.f : pc == 7 ? fromcarry : fromnext;
                carry[ss][k] = gp[k];
            }
            {
                vec_t<T, 8> o;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int e = REV ? 7 - i : i;
                    float a = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) a = fmaf(taps[ss][k], gp[i + 3 - k], a);
                    if (DXACC) a += static_cast<float>(dxold[ss][e]);
                    o[e] = static_cast<T>(a);
                }
                if (emit) {     // the tile after the range contributes nothing (wave-uniform branch, no memory access under it)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        dbacc[ss] += gp[i];
#pragma unroll
                        for (int k = 0; k < 4; ++k) dwacc[ss][k] = fmaf(xv[i + k], gp[i], dwacc[ss][k]);
                    }
                }
                st8(dx_rs, (int64_t)d * p.dx_c_stride, tl, okp && emit, __builtin_bit_cast(s16x8, o));
            }
        }
        // second product: dW_x[r][d] += sum_l dx_dbl[r][l] conv1d_out[d][l] over the tile (same column order in both tiles)
        if (emit) {
            lds_order();
            const lds_s16* const pa = in_lds + i16 * kRowE + 8 * g16;
            const lds_s16* const qa = co + i16 * kRowE + 8 * g16;
#pragma unroll
            for (int s = 0; s < kTL / 32; ++s) {
                const s16x8 bq = *reinterpret_cast<const lds_s16x8*>(qa + 32 * s);
#pragma unroll
                for (int mb = 0; mb < KP / 16; ++mb) {
                    const s16x8 ap = *reinterpret_cast<const lds_s16x8*>(pa + 16 * mb * kRowE + 32 * s);
                    accw[mb] = Mfma16<T>::run(ap, bq, accw[mb]);
                }
            }
        }
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) widen(ss);   // the next tile's inputs (the loads of this iteration's top)
        __syncthreads();          // every wave has read tile t: the dx_dbl buffer may be overwritten
        stage_store();
    }
    // dW_x: one atomic per (r, channel) and workgroup
    {
        const int d = d0 + i16;
        if (d < p.dim) {
#pragma unroll
            for (int mb = 0; mb < KP / 16; ++mb)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int m = 16 * mb + 4 * g16 + v;
                    if (m < R) atomicAdd(p.dw_x + (int64_t)m * p.dwx_k_stride + d, accw[mb][v]);
                }
        }
    }
    // conv dweight / dbias: the 8 lanes of a channel row hold its partial sums
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
        float v[5] = {dwacc[ss][0], dwacc[ss][1], dwacc[ss][2], dwacc[ss][3], dbacc[ss]};
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            v[k] += dpp_mov<DPP_ROW_SHL1, 0xf>(0.f, v[k]);
            v[k] += dpp_mov<DPP_ROW_SHL2, 0xf>(0.f, v[k]);
            v[k] += dpp_mov<DPP_ROW_SHL4, 0xf>(0.f, v[k]);
        }
        const int d = d0 + 8 * ss + rr;
        if (pc == 0 && d < p.dim) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int wi = k - (4 - p.width);
                if (wi >= 0) atomicAdd(p.dconv_weight + (int64_t)d * p.dconv_weight_c_stride + (int64_t)wi * p.dconv_weight_width_stride, v[k]);
            }
            if (p.dconv_bias) atomicAdd(p.dconv_bias + d, v[4]);
        }
    }
}

template <typename T, int KS, bool DXACC, bool RAG>
__global__ __launch_bounds__(kPT, 2) void proj_conv_bwd_kernel(const vms_proj_conv_bwd_params p, const int tiles_per_wg, const int2 p_grid, const int entry) {
    // the batch entry of this workgroup (same decode as in the body) decides the direction: workgroup-uniform
    const int pair = (((int)blockIdx.x >> 3) / p_grid.y) * 8 + ((int)blockIdx.x & 7);
    const int b = pair / p_grid.x;    // >= batch for the padding workgroups of the last group of 8, which return at once
    const bool rev = p.reverse != 0 || (p.reverse_from > 0 && b >= p.reverse_from);
    if (rev) proj_conv_bwd_body<T, KS, true, DXACC, RAG>(p, tiles_per_wg, p_grid, entry);
    else proj_conv_bwd_body<T, KS, false, DXACC, RAG>(p, tiles_per_wg, p_grid, entry);
}

// ---------------------------------------------------------------------------------------------------------------------
// proj_kred:  out[b][m][l] = sum_{k < K} W[m][k] in[b][k][l]     m <= 96 rows, K = the channel count (hundreds .. thousands)
// ---------------------------------------------------------------------------------------------------------------------
// The two skinny products of the inner node whose CONTRACTION runs over the channels: forward  x_dbl = x_proj.weight @ conv1d_out
// (SSI:181: m = dt_rank + 2 d_state), backward  dx_dbl[:R] = dt_proj.weight^T @ ddelta (SSI:276: m = dt_rank).  One pass over the
// (batch, channels, seqlen) activation, an output a tenth of its size: a pure HBM stream with the matrix cores attached, which
// the library's general kernels run at 1.7-2.5x the memory floor for these shapes (profiles/r04_proj_kred.md).
// A workgroup owns TL = 64 / 128 / 256 positions of one batch entry and walks the channels 64 at a time: the activation block
// (64 channels x TL positions, full 128-byte lines) and the weight block (m x 64) go global -> registers -> LDS once per
// workgroup; the 4 waves split the tile 2 (position halves) x 2 (row halves), the activation block becomes the MFMA B operand
// through ds_read_b64_tr_b16 (the channels are its STRIDED axis), the weight block the A operand -- by 16-byte reads when it is
// stored (m, k) with k contiguous (x_proj.weight), by transposing reads when it is stored (k, m) (dt_proj.weight used as its
// transpose: WT).  fp32 accumulators (16x16x32 MFMA) stay in registers over the whole channel loop; results leave as 16-bit rows.
// A second problem of the same shape (w2 / in2 / out2: the other direction of a bidirectional block) rides in the same grid.
template <typename T, int MH, int NL, bool WT>
__global__ __launch_bounds__(kPT, (NL <= 2 ? 4 : NL <= 4 ? 3 : 2)) void proj_kred_kernel(const vms_proj_kred_params p) {
    constexpr int TL = 32 * NL;                       // positions per workgroup: 2 halves x NL 16-wide blocks
    constexpr int KB = 64;                            // channels per step
    constexpr int INP = TL == 64 ? 72 : TL + 16;      // pitch of the activation block, elements: 4 consecutive rows x 32 bytes on distinct banks
    constexpr int MP = 32 * MH;                       // rows padded: 2 halves x MH 16-row blocks
    constexpr int WP = WT ? MP + 8 : KB + 8;          // pitch of the weight block, elements
    constexpr int PPR = TL / 8, RPP = kPT / PPR;      // activation block: 16-byte pieces per row, rows per pass of the workgroup (NL passes)
    constexpr int PW = MP / 8;                        // WT: pieces per k row of the weight block
    typedef __attribute__((address_space(3))) short lds_s16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, g16 = lane >> 4;
    const int wl = wave & 1, wm = wave >> 1;
    const int dir = (int)blockIdx.z >= p.batch ? 1 : 0, b = (int)blockIdx.z - dir * p.batch;
    const int L = p.seqlen, K = p.k, M = p.m;
    const int l0 = blockIdx.x * TL;
    const T* const in_b = static_cast<const T*>(dir ? p.in2 : p.in) + (int64_t)b * p.in_batch_stride;
    T* const out_b = static_cast<T*>(dir ? p.out2 : p.out) + (int64_t)b * p.out_batch_stride;
    lds_s16* const in_lds = (lds_s16*)reinterpret_cast<short*>(smem);                      // [KB][INP]
    lds_s16* const w_lds = in_lds + KB * INP;                                              // [MP][WP] or, WT, [KB][WP]

    // Every load of the loop goes through a buffer resource with an out-of-range offset for what must not be read (channels
    // beyond K, positions beyond seqlen, blocks beyond the last): such loads return 0 and touch no memory, so no load sits under a
    // branch and the compiler counts them -- the blocks two steps ahead stay in flight across the barriers (s_waitcnt vmcnt(N)).
    const __amdgpu_buffer_rsrc_t in_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(in_b), 0, (int)(((int64_t)(K - 1) * p.in_k_stride + L) * 2), kPBufFlags);
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(static_cast<const T*>(dir ? p.w2 : p.w)), 0, (int)(((int64_t)(M - 1) * p.w_row_stride + (int64_t)(K - 1) * p.w_k_stride + 1) * 2), kPBufFlags);
    constexpr int kOOB = -1;
    struct Stage { s16x8 i[NL], w[MH]; };
    const int ipc = tid % PPR, irow = tid / PPR;
    auto stage_load = [&](Stage& st, int kb) __attribute__((always_inline)) {
        const int l = l0 + 8 * ipc;
#pragma unroll
        for (int ps = 0; ps < NL; ++ps) {
            const int k = kb * KB + irow + RPP * ps;
            st.i[ps] = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(in_rs, k < K && l < L ? (int)(((int64_t)k * p.in_k_stride + l) * 2) : kOOB, 0, 0));
        }
#pragma unroll
        for (int ps = 0; ps < MH; ++ps) {
            if (WT) {     // row k of the block = 8 PW consecutive m
                const int idx = tid + kPT * ps, rk = idx / PW, m = 8 * (idx % PW), k = kb * KB + rk;
                st.w[ps] = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(w_rs, k < K && m < M ? (int)(((int64_t)k * p.w_k_stride + m) * 2) : kOOB, 0, 0));
            } else {      // row m of the block = 64 consecutive k
                const int m = (tid >> 3) + 32 * ps, k = kb * KB + 8 * (tid & 7);
                st.w[ps] = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(w_rs, m < M && k < K ? (int)(((int64_t)m * p.w_row_stride + k) * 2) : kOOB, 0, 0));
            }
        }
    };
    auto stage_store = [&](const Stage& st) __attribute__((always_inline)) {
#pragma unroll
        for (int ps = 0; ps < NL; ++ps)
            *reinterpret_cast<lds_s16x8*>(in_lds + (irow + RPP * ps) * INP + 8 * ipc) = st.i[ps];
#pragma unroll
        for (int ps = 0; ps < MH; ++ps) {
            if (WT) {
                const int idx = tid + kPT * ps;
                *reinterpret_cast<lds_s16x8*>(w_lds + (idx / PW) * WP + 8 * (idx % PW)) = st.w[ps];
            } else {
                *reinterpret_cast<lds_s16x8*>(w_lds + ((tid >> 3) + 32 * ps) * WP + 8 * (tid & 7)) = st.w[ps];
            }
        }
    };

    f32x4 acc[MH][NL];
#pragma unroll
    for (int mh = 0; mh < MH; ++mh)
#pragma unroll
        for (int nl = 0; nl < NL; ++nl) acc[mh][nl] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transposing reads (see proj_apply): lane i of a 16-lane group supplies row i / 4, columns 4 (i % 4) .. + 3 of a [4][16] block and
    // receives column i of its 4 rows.  16x16x32 operands: lane -> (row / column lane & 15, k = 8 (lane >> 4) .. + 7)
    const lds_s16* const tb = in_lds + (8 * g16 + (i16 >> 2)) * INP + wl * (TL / 2) + 4 * (i16 & 3);
    const int m0w = wm * 16 * MH;
    const lds_s16* const ta = WT ? w_lds + (8 * g16 + (i16 >> 2)) * WP + m0w + 4 * (i16 & 3) : w_lds + (m0w + i16) * WP + 8 * g16;

    auto compute = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < KB / 32; ++s) {
            s16x8 af[MH];
#pragma unroll
            for (int mh = 0; mh < MH; ++mh) {
                if (WT) {
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ta + (32 * s) * WP + 16 * mh));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ta + (32 * s + 4) * WP + 16 * mh));
                    af[mh] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                } else {
                    af[mh] = *reinterpret_cast<const lds_s16x8*>(ta + 16 * mh * WP + 32 * s);
                }
            }
#pragma unroll
            for (int nl = 0; nl < NL; ++nl) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tb + (32 * s) * INP + 16 * nl));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tb + (32 * s + 4) * INP + 16 * nl));
                const s16x8 bf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int mh = 0; mh < MH; ++mh) acc[mh][nl] = Mfma16<T>::run(af[mh], bf, acc[mh][nl]);
            }
        }
    };

    const int nkb = (K + KB - 1) / KB;
    Stage sa, sb;     // blocks kb and kb + 1: two steps of requests in flight
    stage_load(sa, 0);
    stage_load(sb, 1);
    for (int kb = 0; kb < nkb; kb += 2) {
        stage_store(sa);
        __syncthreads();          // blocks kb are in LDS
        stage_load(sa, kb + 2);
        compute();
        __syncthreads();          // every wave has read blocks kb
        if (kb + 1 < nkb) {       // (workgroup-uniform; no load under it)
            stage_store(sb);
            __syncthreads();
        }
        stage_load(sb, kb + 3);
        if (kb + 1 < nkb) {
            compute();
            __syncthreads();
        }
    }
    // C layout: column = lane & 15 (position), row = 4 (lane >> 4) + register (output row)
#pragma unroll
    for (int mh = 0; mh < MH; ++mh)
#pragma unroll
        for (int nl = 0; nl < NL; ++nl)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int m = m0w + 16 * mh + 4 * g16 + v, l = l0 + wl * (TL / 2) + 16 * nl + i16;
                if (m < M && l < L) out_b[(int64_t)m * p.out_row_stride + l] = static_cast<T>(acc[mh][nl][v]);
            }
    // side job: the rows that follow the product in `out` <- the scan backward's fp32 dB / dC sums of this tile's positions
    const float* const cs = dir ? p.cast_src2 : p.cast_src;
    if (cs) {
        const int n_rows = p.cast_rows * p.cast_groups;
        for (int idx = tid; idx < n_rows * (TL / 4); idx += kPT) {
            const int r = idx / (TL / 4), l = l0 + 4 * (idx % (TL / 4));
            if (l < L) {
                const int g = r / p.cast_rows, n = r - g * p.cast_rows;
                const f32x4 v = *reinterpret_cast<const f32x4*>(cs + (int64_t)g * p.cast_group_stride + (int64_t)b * p.cast_batch_stride + (int64_t)n * p.cast_row_stride + l);
                vec_t<T, 4> o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = static_cast<T>(v[e]);
                *reinterpret_cast<vec_t<T, 4>*>(out_b + (int64_t)(M + r) * p.out_row_stride + l) = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// conv_xproj_dual: both directions' causal_conv1d (+ SiLU) of a bidirectional block AND both x_proj products, one pass over x
// ---------------------------------------------------------------------------------------------------------------------
// The forward of a ViM block runs  conv1d_out = silu(conv1d(x)) (SSI:170-176) and x_dbl = x_proj.weight @ conv1d_out (:181) for
// each direction: vms_causal_conv1d_fwd_dual writes the two conv1d outputs from one pass over x and vms_proj_kred reads both
// back for the two products.  Here the workgroup layout of proj_kred (TL positions of one batch entry, channels 64 at a time) also
// computes the convolutions: a thread's 16-byte piece of x (8 positions of one channel, neighbours' halos through DPP, the
// tile's edges from memory) becomes the piece of BOTH conv1d outputs -- the arithmetic of conv_fwd_dual_kernel, tap for tap, so
// the outputs are bit-identical -- which it stores to HBM and into the two LDS tiles the matrix cores then read as B operands.
// conv1d_out is never read back: 402 MB instead of 671 at (8, 1024, 8192), one launch instead of two.
template <typename T, typename WT_, int MH, int NL>
__global__ __launch_bounds__(kPT, 2) void conv_xproj_dual_kernel(const vms_conv_xproj_dual_params q, const int entry) {
    const vms_conv_fwd_params& p = q.c.f;
    constexpr int TL = 32 * NL, KB = 64, INP = TL == 64 ? 72 : TL + 16, MP = 32 * MH, WP = KB + 8;
    constexpr int PPR = TL / 8, RPP = kPT / PPR;
    constexpr int TPP = 8;                            // floats per (direction, channel) record of conv taps: 4 taps, bias, 3 unused
    typedef __attribute__((address_space(3))) short lds_s16;
    typedef unsigned int u32;
    typedef u32 u32x2 __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, g16 = lane >> 4;
    const int wl = wave & 1, wm = wave >> 1;
    const int L = p.seqlen, K = p.dim, M = q.m;
    // 1-D grid; workgroup id = position tile fastest (what the (tiles, 1, batch) grid of round 4 dispatched).  A tile reads 6 bytes of
    // either neighbour's 128-byte line per channel row (the convolutions' halos); neighbouring tiles sit on different XCDs (id % 8),
    // every L2 fetches those lines for itself and FETCH_SIZE reads 412 MB for 134 MB of x.  VMS_CXP_XCD=1 decodes the id per XCD
    // instead (XCD x runs the x-th eighth of the (batch entry, tile) list in order): FETCH_SIZE 149 MB -- and 139-141 us instead of
    // 132-134 on the same box (profiles/r05_fused_traffic.md): the re-fetched lines come out of the Infinity Cache, the kernel is
    // not bound by them, and one batch entry per XCD walks the channels in lock-step on one L2.  Measured, not adopted.
    const int n_tl = (L + TL - 1) / TL, n_wg = n_tl * p.batch, per_xcd = (n_wg + 7) >> 3;
#ifndef VMS_CXP_XCD
#define VMS_CXP_XCD 0   /* 1 (A/B builds): neighbouring position tiles on one XCD (see above) */
#endif
// timing-only ablations (wrong results), round 6: tools/variant.sh <tag> -DVMS_ABL_CXP_NOBAR=1 / _NOSTORE=1 / _NOCONV=1 / _NOMMA=1 / _NOHALO=1
#ifndef VMS_ABL_CXP_NOBAR
#define VMS_ABL_CXP_NOBAR 0
#endif
#ifndef VMS_ABL_CXP_NOSTORE
#define VMS_ABL_CXP_NOSTORE 0
#endif
#ifndef VMS_ABL_CXP_NOCONV
#define VMS_ABL_CXP_NOCONV 0
#endif
#ifndef VMS_ABL_CXP_NOMMA
#define VMS_ABL_CXP_NOMMA 0
#endif
#ifndef VMS_ABL_CXP_NOHALO
#define VMS_ABL_CXP_NOHALO 0
#endif
    const int lid = VMS_CXP_XCD ? ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    if (lid >= n_wg) return;
    const int b = lid / n_tl;
    const int l0 = (lid - b * n_tl) * TL;
    lds_s16* const in_a = (lds_s16*)reinterpret_cast<short*>(smem);                        // [KB][INP] conv1d_out, left-to-right set
    lds_s16* const in_b = in_a + KB * INP;                                                 // [KB][INP] right-to-left set
    lds_s16* const w_a = in_b + KB * INP;                                                  // [MP][WP] x_proj.weight block
    lds_s16* const w_b = w_a + MP * WP;
    lds_f32* const tp = (lds_f32*)reinterpret_cast<float*>(smem + (2 * KB * INP + 2 * MP * WP) * 2);   // [2 buffers][2 directions][KB][TPP]

    const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(static_cast<const T*>(p.x) + (int64_t)b * p.x_batch_stride), 0, (int)(((int64_t)(K - 1) * p.x_c_stride + L) * 2), kPBufFlags);
    const __amdgpu_buffer_rsrc_t oa_rs = __builtin_amdgcn_make_buffer_rsrc(
        static_cast<T*>(p.out) + (int64_t)b * p.out_batch_stride, 0, (int)(((int64_t)(K - 1) * p.out_c_stride + L) * 2), kPBufFlags);
    const __amdgpu_buffer_rsrc_t ob_rs = __builtin_amdgcn_make_buffer_rsrc(
        static_cast<T*>(q.c.out_b) + (int64_t)b * q.c.out_b_batch_stride, 0, (int)(((int64_t)(K - 1) * q.c.out_b_c_stride + L) * 2), kPBufFlags);
    const int w_bytes = (int)(((int64_t)(M - 1) * q.wx_row_stride + K) * 2);
    const __amdgpu_buffer_rsrc_t wa_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(q.w_x), 0, w_bytes, kPBufFlags);
    const __amdgpu_buffer_rsrc_t wb_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(q.w_x_b), 0, w_bytes, kPBufFlags);
    // conv taps / biases: thread t < 128 owns (direction t >> 6, channel t & 63) of the block; a missing bias is a resource of size 0
    const int t_dir = wave & 1, t_ch = tid & 63;     // (from the wave index, a scalar: the two resources below stay in SGPRs -- built from a
                                                     // per-thread value they are "divergent" and every load through them becomes a waterfall loop)
    const int64_t cw_c = t_dir ? q.c.weight_b_c_stride : p.weight_c_stride, cw_w = t_dir ? q.c.weight_b_width_stride : p.weight_width_stride;
    const __amdgpu_buffer_rsrc_t cw_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(t_dir ? q.c.weight_b : p.weight), 0, (int)(((int64_t)(K - 1) * cw_c + (int64_t)(p.width - 1) * cw_w + 1) * sizeof(WT_)), kPBufFlags);
    const void* const cbp = t_dir ? q.c.bias_b : p.bias;
    const __amdgpu_buffer_rsrc_t cb_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(cbp), 0, cbp ? K * (int)sizeof(WT_) : 0, kPBufFlags);
    constexpr int kOOB = -1;

    struct XStage { s16x8 m[NL]; u32x2 el[NL], er[NL]; };     // a thread's pieces of x: 8 positions + what precedes / follows the tile's edge
    struct WStage { s16x8 wa[MH], wb[MH]; float tp[5]; };
    const int ipc = tid % PPR, irow = tid / PPR;
    auto x_load = [&](XStage& st, int kb) __attribute__((always_inline)) {
        const int l = l0 + 8 * ipc;
#pragma unroll
        for (int ps = 0; ps < NL; ++ps) {
            const int k = kb * KB + irow + RPP * ps;
            const bool ok = k < K && l < L;
            const int off = (int)(((int64_t)k * p.x_c_stride + l) * 2);
            st.m[ps] = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(x_rs, ok ? off : kOOB, 0, 0));
            // positions l - 4 .. l - 1 (the tile's first piece) and l + 8 .. l + 11 (its last): zeros outside the row
            st.el[ps] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(x_rs, !VMS_ABL_CXP_NOHALO && ok && ipc == 0 && l > 0 ? off - 8 : kOOB, 0, 0));
            st.er[ps] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(x_rs, !VMS_ABL_CXP_NOHALO && ok && ipc == PPR - 1 && l + 8 < L ? off + 16 : kOOB, 0, 0));
        }
    };
    auto tap_val = [&](int off_elems, const __amdgpu_buffer_rsrc_t& rs, bool ok) __attribute__((always_inline)) -> float {
        if constexpr (sizeof(WT_) == 4) {
            return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, ok ? off_elems * 4 : kOOB, 0, 0));
        } else {
            const unsigned short h = __builtin_amdgcn_raw_buffer_load_b16(rs, ok ? off_elems * 2 : kOOB, 0, 0);
            return static_cast<float>(__builtin_bit_cast(WT_, h));
        }
    };
    // W blocks of step kb, conv taps of step kb_t
    auto w_load = [&](WStage& st, int kb, int kb_t) __attribute__((always_inline)) {
#pragma unroll
        for (int ps = 0; ps < MH; ++ps) {
            const int m = (tid >> 3) + 32 * ps, k = kb * KB + 8 * (tid & 7);
            const int off = m < M && k < K ? (int)(((int64_t)m * q.wx_row_stride + k) * 2) : kOOB;
            st.wa[ps] = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(wa_rs, off, 0, 0));
            st.wb[ps] = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(wb_rs, off, 0, 0));
        }
        const int c = kb_t * KB + t_ch;
        const bool okc = tid < 2 * KB && c < K;
#pragma unroll
        for (int i = 0; i < 4; ++i) {       // taps[i] multiplies x[l - 3 + i] (left-to-right) / x[l + 3 - i] (right-to-left); widths < 4: leading zeros
            const int wi = i - (4 - p.width);
            st.tp[i] = tap_val((int)((int64_t)c * cw_c + (int64_t)wi * cw_w), cw_rs, okc && wi >= 0);
        }
        st.tp[4] = tap_val(c, cb_rs, okc);
    };
    auto w_store = [&](const WStage& st, int tbuf) __attribute__((always_inline)) {
#pragma unroll
        for (int ps = 0; ps < MH; ++ps) {
            *reinterpret_cast<lds_s16x8*>(w_a + ((tid >> 3) + 32 * ps) * WP + 8 * (tid & 7)) = st.wa[ps];
            *reinterpret_cast<lds_s16x8*>(w_b + ((tid >> 3) + 32 * ps) * WP + 8 * (tid & 7)) = st.wb[ps];
        }
        if (tid < 2 * KB) {
            lds_f32* dst = tp + ((tbuf * 2 + t_dir) * KB + t_ch) * TPP;
            *(lds_f32x4*)dst = f32x4{st.tp[0], st.tp[1], st.tp[2], st.tp[3]};
            dst[4] = st.tp[4];
        }
    };
    auto up = [](u32 d, int hi) __attribute__((always_inline)) -> float {    // element `hi` of a dword of two T
        return static_cast<float>(__builtin_bit_cast(T, (unsigned short)(hi ? d >> 16 : d & 0xffffu)));
    };
    // the convolutions of step kb's pieces: both conv1d outputs to HBM and to the LDS tiles
    auto conv_store = [&](const XStage& st, int kb, int tbuf) __attribute__((always_inline)) {
        const int l = l0 + 8 * ipc;
#pragma unroll
        for (int ps = 0; ps < NL; ++ps) {
            const int r = irow + RPP * ps, k = kb * KB + r;
            typedef u32 u32x4_ __attribute__((ext_vector_type(4)));
            const u32x4_ md = __builtin_bit_cast(u32x4_, st.m[ps]);
            // the previous piece's positions 5, 6, 7 and the next piece's 0, 1, 2 sit in the neighbouring lanes
            u32 pl2 = (u32)__builtin_amdgcn_update_dpp(0, (int)md[2], DPP_WAVE_SHR1, 0xf, 0xf, false);
            u32 pl3 = (u32)__builtin_amdgcn_update_dpp(0, (int)md[3], DPP_WAVE_SHR1, 0xf, 0xf, false);
            u32 pr0 = (u32)__builtin_amdgcn_update_dpp(0, (int)md[0], DPP_WAVE_SHL1, 0xf, 0xf, false);
            u32 pr1 = (u32)__builtin_amdgcn_update_dpp(0, (int)md[1], DPP_WAVE_SHL1, 0xf, 0xf, false);
            if (ipc == 0) { pl2 = st.el[ps][0]; pl3 = st.el[ps][1]; }
            if (ipc == PPR - 1) { pr0 = st.er[ps][0]; pr1 = st.er[ps][1]; }
            // entry (> 0: 8 or 16): the row is `seqlen / entry` independent sequences laid end to end (vms_conv_xproj_dual folds gap-free
            // batches of short sequences): no filter reaches across their boundaries
            if (entry != 0 && (l & (entry - 1)) == 0) { pl2 = 0; pl3 = 0; }
            if (entry != 0 && ((l + 8) & (entry - 1)) == 0) { pr0 = 0; pr1 = 0; }
            float xv[8 + 6];     // xv[3 + i] = x[l + i]; [0..2] = the 3 positions before, [11..13] = the 3 after
            xv[0] = up(pl2, 1); xv[1] = up(pl3, 0); xv[2] = up(pl3, 1);
#pragma unroll
            for (int i = 0; i < 8; ++i) xv[3 + i] = up(md[i >> 1], i & 1);
            xv[11] = up(pr0, 0); xv[12] = up(pr0, 1); xv[13] = up(pr1, 0);
            const lds_f32* ta_ = tp + ((tbuf * 2 + 0) * KB + r) * TPP;
            const lds_f32* tb_ = tp + ((tbuf * 2 + 1) * KB + r) * TPP;
            const f32x4 tav = *(const lds_f32x4*)ta_, tbv = *(const lds_f32x4*)tb_;
            const float bias_a = ta_[4], bias_b = tb_[4];
            const float taps_a[4] = {tav.x, tav.y, tav.z, tav.w}, taps_b[4] = {tbv.x, tbv.y, tbv.z, tbv.w};
            vec_t<T, 8> oa, ob;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float acc = bias_a, acc_b = bias_b;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    acc = fmaf(taps_a[kk], xv[i + kk], acc);
                    acc_b = fmaf(taps_b[kk], xv[i + 6 - kk], acc_b);
                }
                float ra = VMS_ABL_CXP_NOCONV ? acc : acc * sigmoidf_(acc), rb = VMS_ABL_CXP_NOCONV ? acc_b : acc_b * sigmoidf_(acc_b);
                asm volatile("" : "+v"(ra), "+v"(rb));   // as conv_fwd_dual_kernel: narrowed from the rounded fp32 value
                oa[i] = static_cast<T>(ra);
                ob[i] = static_cast<T>(rb);
            }
            const bool ok = k < K && l < L && !VMS_ABL_CXP_NOSTORE;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pu32x4, oa), oa_rs, ok ? (int)(((int64_t)k * p.out_c_stride + l) * 2) : kOOB, 0, VMS_CXP_ST_AUX);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pu32x4, ob), ob_rs, ok ? (int)(((int64_t)k * q.c.out_b_c_stride + l) * 2) : kOOB, 0, VMS_CXP_ST_AUX);
            *reinterpret_cast<lds_s16x8*>(in_a + r * INP + 8 * ipc) = __builtin_bit_cast(s16x8, oa);
            *reinterpret_cast<lds_s16x8*>(in_b + r * INP + 8 * ipc) = __builtin_bit_cast(s16x8, ob);
        }
    };

    f32x4 acc_a[MH][NL], acc_b[MH][NL];
#pragma unroll
    for (int mh = 0; mh < MH; ++mh)
#pragma unroll
        for (int nl = 0; nl < NL; ++nl) { acc_a[mh][nl] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_b[mh][nl] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int m0w = wm * 16 * MH;
    const int tb_off = (8 * g16 + (i16 >> 2)) * INP + wl * (TL / 2) + 4 * (i16 & 3);
    const int ta_off = (m0w + i16) * WP + 8 * g16;
    auto product = [&](const lds_s16* in_t, const lds_s16* w_t, f32x4 (&acc)[MH][NL]) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < KB / 32; ++s) {
            s16x8 af[MH];
#pragma unroll
            for (int mh = 0; mh < MH; ++mh) af[mh] = *reinterpret_cast<const lds_s16x8*>(w_t + ta_off + 16 * mh * WP + 32 * s);
#pragma unroll
            for (int nl = 0; nl < NL; ++nl) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(in_t + tb_off + (32 * s) * INP + 16 * nl));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(in_t + tb_off + (32 * s + 4) * INP + 16 * nl));
                const s16x8 bf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int mh = 0; mh < MH; ++mh) acc[mh][nl] = Mfma16<T>::run(af[mh], bf, acc[mh][nl]);
            }
        }
    };

    const int nkb = (K + KB - 1) / KB;
    XStage sa, sb;      // x of steps kb and kb + 1 (HBM latency: two steps of requests in flight)
    WStage sw;          // W_x blocks of step kb, conv taps of step kb + 1 (L2 latency: one step)
    w_load(sw, 0, 0);
    x_load(sa, 0);
    w_store(sw, 0);     // taps of step 0 (the W blocks are stored again below)
    w_load(sw, 0, 1);
    x_load(sb, 1);
    __syncthreads();
    auto step = [&](XStage& st, int kb) __attribute__((always_inline)) {
        conv_store(st, kb, kb & 1);
        w_store(sw, (kb + 1) & 1);      // W blocks of this step, taps of the next
        // (requests return in issue order: the L2-resident W blocks go out BEFORE the x pieces of two steps ahead, so that waiting for
        // them at the next step's w_store leaves those x requests in flight)
        w_load(sw, kb + 1, kb + 2);
        x_load(st, kb + 2);
        if (!VMS_ABL_CXP_NOBAR) __syncthreads();                // both conv1d tiles and the W blocks are in LDS
        if (!VMS_ABL_CXP_NOMMA) {
            product(in_a, w_a, acc_a);
            product(in_b, w_b, acc_b);
        }
        if (!VMS_ABL_CXP_NOBAR) __syncthreads();                // every wave has read them
    };
    for (int kb = 0; kb < nkb; kb += 2) {
        step(sa, kb);
        if (kb + 1 < nkb) step(sb, kb + 1);    // (workgroup-uniform)
    }
    T* const xa_b = static_cast<T*>(q.x_dbl) + (int64_t)b * q.xdbl_batch_stride;
    T* const xb_b = static_cast<T*>(q.x_dbl_b) + (int64_t)b * q.xdbl_batch_stride;
#pragma unroll
    for (int mh = 0; mh < MH; ++mh)
#pragma unroll
        for (int nl = 0; nl < NL; ++nl)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int m = m0w + 16 * mh + 4 * g16 + v, l = l0 + wl * (TL / 2) + 16 * nl + i16;
                if (m < M && l < L) {
                    xa_b[(int64_t)m * q.xdbl_row_stride + l] = static_cast<T>(acc_a[mh][nl][v]);
                    xb_b[(int64_t)m * q.xdbl_row_stride + l] = static_cast<T>(acc_b[mh][nl][v]);
                }
            }
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------

template <typename T, int KS>
static int launch_apply(const vms_proj_apply_params& p, hipStream_t stream) {
    const int n_tiles = (p.seqlen + kTL - 1) / kTL;
    const int d_tiles = (p.rows + 127) / 128;
    // ~2 workgroups per CU, at least 4 tiles each (the W fragments are gathered once per workgroup): (8, 1024, 8192) k = 64
    // 33 us with 16 tiles per workgroup, 38 us with 8, 47 us with 4, 56 us with 32 (tools/kb_proj.py sweep)
    const int64_t want = 2 * (int64_t)device_cu_count();
    int tpw = (int)(((int64_t)n_tiles * d_tiles * p.batch + want - 1) / want);
    if (tpw < 4) tpw = 4;
    if (p.tiles_per_wg > 0) tpw = p.tiles_per_wg;
    if (tpw > n_tiles) tpw = n_tiles;
    const dim3 grid((n_tiles + tpw - 1) / tpw, d_tiles, p.batch), block(kPT);
    const size_t smem = (size_t)KS * 16 * kRowE * 2 + (size_t)4 * 32 * kEpE * sizeof(float);
    if (p.accumulate) hipLaunchKernelGGL((proj_apply_kernel<T, KS, true>), grid, block, smem, stream, p, tpw);
    else hipLaunchKernelGGL((proj_apply_kernel<T, KS, false>), grid, block, smem, stream, p, tpw);
    VMS_LAUNCH_CHECK();
    set_last_kernel(p.accumulate ? "proj_apply+acc" : "proj_apply");
    return VMS_OK;
}

template <typename T>
static int dispatch_apply(const vms_proj_apply_params& p, hipStream_t stream) {
    switch ((p.k + 15) / 16) {
        case 1: return launch_apply<T, 1>(p, stream);
        case 2: return launch_apply<T, 2>(p, stream);
        case 3: return launch_apply<T, 3>(p, stream);
        case 4: return launch_apply<T, 4>(p, stream);
        case 5: return launch_apply<T, 5>(p, stream);
        default: return launch_apply<T, 6>(p, stream);
    }
}

template <typename T, int MB>
static int launch_wgrad(const vms_proj_wgrad_params& p, hipStream_t stream) {
    const int n_tiles = (p.seqlen + kTL - 1) / kTL;
    const int n_blocks = (p.n + 127) / 128;
    // ~2 workgroups per CU; a workgroup ends with m x 128 atomics, so keep its range >= 8 tiles (512 positions) when the row
    // allows (16 until round 4: right at (8, 1024, 8192), but (8, 768, 3136) then ran 192 workgroups: 22.8 us, 17.8 with 8; (4, 512,
    // 2304) 11.7 -> 7.4 us; profiles/r04_small_proj_sweep.txt)
    const int64_t want = 2 * (int64_t)device_cu_count();
    int tpw = (int)(((int64_t)n_tiles * n_blocks * p.batch + want - 1) / want);
    if (tpw < 8) tpw = 8;
    if (p.tiles_per_wg > 0) tpw = p.tiles_per_wg;
    if (tpw > n_tiles) tpw = n_tiles;
    const dim3 grid((n_tiles + tpw - 1) / tpw, n_blocks, p.batch), block(kPT);
    const size_t smem = (size_t)MB * 32 * kRowE * 2 + (size_t)4 * 32 * kRowE * 2;
    hipLaunchKernelGGL((proj_wgrad_kernel<T, MB>), grid, block, smem, stream, p, tpw);
    VMS_LAUNCH_CHECK();
    set_last_kernel("proj_wgrad");
    return VMS_OK;
}

template <typename T>
static int dispatch_wgrad(const vms_proj_wgrad_params& p, hipStream_t stream) {
    switch ((p.m + 31) / 32) {
        case 1: return launch_wgrad<T, 1>(p, stream);
        case 2: return launch_wgrad<T, 2>(p, stream);
        case 3: return launch_wgrad<T, 3>(p, stream);
        default: return launch_wgrad<T, 4>(p, stream);
    }
}

template <typename T, int MH, int NL, bool WT>
static int launch_kred(const vms_proj_kred_params& p, hipStream_t stream) {
    constexpr int TL = 32 * NL, KB = 64, INP = TL == 64 ? 72 : TL + 16, MP = 32 * MH, WP = WT ? MP + 8 : KB + 8;
    const int ndir = p.in2 ? 2 : 1;
    const dim3 grid((p.seqlen + TL - 1) / TL, 1, p.batch * ndir), block(kPT);
    const size_t smem = (size_t)KB * INP * 2 + (size_t)(WT ? KB : MP) * WP * 2;
    hipLaunchKernelGGL((proj_kred_kernel<T, MH, NL, WT>), grid, block, smem, stream, p);
    VMS_LAUNCH_CHECK();
    set_last_kernel(ndir == 2 ? (WT ? "proj_kred_t+dual" : "proj_kred+dual") : (WT ? "proj_kred_t" : "proj_kred"));
    return VMS_OK;
}

template <typename T, int MH, bool WT>
static int dispatch_kred_tile(const vms_proj_kred_params& p, hipStream_t stream) {
    // 128 positions per workgroup once that still gives every CU 1.5 workgroups, else 64 (the weight block is re-read from L2 once
    // per tile: 1.5x the activation's bytes at 64 positions, 0.75x at 128); 256 measured slower than 128 on every grid that allows
    // it (one workgroup per CU: (8, 1024, 8192) both directions 59 vs 53 us; profiles/r04_proj_kred.md) and is kept as a knob (`tile`)
    const int ndir = p.in2 ? 2 : 1;
    int tile = p.tile;
    if (tile != 64 && tile != 128 && tile != 256)
        tile = 2 * (int64_t)p.batch * ndir * ((p.seqlen + 127) / 128) >= 3 * (int64_t)device_cu_count() ? 128 : 64;
    switch (tile) {
        case 256: return launch_kred<T, MH, 8, WT>(p, stream);
        case 128: return launch_kred<T, MH, 4, WT>(p, stream);
        default: return launch_kred<T, MH, 2, WT>(p, stream);
    }
}

template <typename T>
static int dispatch_kred(const vms_proj_kred_params& p, hipStream_t stream) {
    const bool wt = p.w_k_stride != 1;
    const int mh = ((p.m + 15) / 16 + 1) / 2;
#define VMS_KRED(MH_) (wt ? dispatch_kred_tile<T, MH_, true>(p, stream) : dispatch_kred_tile<T, MH_, false>(p, stream))
    switch (mh) {
        case 1: return VMS_KRED(1);
        case 2: return VMS_KRED(2);
        default: return VMS_KRED(3);
    }
#undef VMS_KRED
}

template <typename T, typename WT_, int MH, int NL>
static int launch_conv_xproj(const vms_conv_xproj_dual_params& q, hipStream_t stream, const int entry) {
    constexpr int TL = 32 * NL, KB = 64, INP = TL == 64 ? 72 : TL + 16, MP = 32 * MH, WP = KB + 8;
    const int n_wg = ((q.c.f.seqlen + TL - 1) / TL) * q.c.f.batch;
    const dim3 grid(8 * ((n_wg + 7) / 8)), block(kPT);   // whole eighths: the kernel decodes (XCD, index on it)
    const size_t smem = (size_t)(2 * KB * INP + 2 * MP * WP) * 2 + (size_t)2 * 2 * KB * 8 * sizeof(float);
    if (smem > 64 * 1024) {
        static PerDeviceOnce attr_once;
        const hipError_t arc = attr_once.run([&]() -> hipError_t {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_xproj_dual_kernel<T, WT_, MH, NL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        });
        if (arc != hipSuccess) {
            set_error("hipFuncSetAttribute(conv_xproj_dual, %zu bytes of LDS) failed: %s", smem, hipGetErrorString(arc));
            return VMS_ERR_LAUNCH;
        }
    }
    hipLaunchKernelGGL((conv_xproj_dual_kernel<T, WT_, MH, NL>), grid, block, smem, stream, q, entry);
    VMS_LAUNCH_CHECK();
    set_last_kernel(NL == 4 ? "conv_xproj_dual_128" : "conv_xproj_dual_64");
    return VMS_OK;
}

template <typename T, typename WT_>
static int dispatch_conv_xproj(const vms_conv_xproj_dual_params& q, hipStream_t stream, const int entry) {
    // 64 positions per workgroup: two workgroups per CU; 128 (one per CU, 256 registers) measured 25-45 % slower on every shape
    // (profiles/r04_conv_xproj.md) and stays a knob
    const int tile = q.tile == 128 ? 128 : 64;
    const int mh = ((q.m + 15) / 16 + 1) / 2;
#define VMS_CXP(MH_) (tile == 128 ? launch_conv_xproj<T, WT_, MH_, 4>(q, stream, entry) : launch_conv_xproj<T, WT_, MH_, 2>(q, stream, entry))
    switch (mh) {
        case 1: return VMS_CXP(1);
        case 2: return VMS_CXP(2);
        default: return VMS_CXP(3);
    }
#undef VMS_CXP
}

template <typename T, int KS>
static int launch_conv_bwd(const vms_proj_conv_bwd_params& p, hipStream_t stream, const int entry) {
    constexpr int MB = (KS * 16 + 31) / 32;
    const int n_tiles = (p.seqlen + kTL - 1) / kTL;
    const int d_tiles = (p.dim + kCD - 1) / kCD;
    // ~2 workgroups per CU (2 are resident); every workgroup pays one extra tile (the carry) and 64 x k atomics at its end:
    // (8, 1024, 8192) 113 us with 32 tiles per workgroup, 119 with 16, 132 with 8, 144 with 64 (profiles/r03w_kb_proj.txt)
    const int64_t want = 2 * (int64_t)device_cu_count();
    int tpw = (int)(((int64_t)n_tiles * d_tiles * p.batch + want - 1) / want);
    if (tpw < 8) tpw = 8;
    if (tpw > n_tiles) tpw = n_tiles;
    // two workgroups are resident per CU: a grid a little above 2 per CU -- the padding to whole XCD groups can push it there --
    // runs a second, nearly empty round ((1, 768, 65536): 24 tiles per workgroup = 576 workgroups 125 us, 16 = 768 or 32 = 384
    // 93-98 us): longer ranges until the grid fits one round
    auto grid_of = [&](int t) { return 8 * (int64_t)((((n_tiles + t - 1) / t) * p.batch + 7) / 8) * d_tiles; };
    while (tpw < n_tiles && grid_of(tpw) > want && 2 * grid_of(tpw) < 3 * want) ++tpw;
    if (p.tiles_per_wg > 0) tpw = p.tiles_per_wg;
#ifdef VMS_PCB_TPW
    tpw = VMS_PCB_TPW;            /* A/B builds: tiles per workgroup of vms_proj_conv_bwd */
#endif
    if (tpw > n_tiles) tpw = n_tiles;
    const int n_rng = (n_tiles + tpw - 1) / tpw;
    const int2 pg = make_int2(n_rng, d_tiles);
    const dim3 grid(8 * ((n_rng * p.batch + 7) / 8) * d_tiles), block(kPT);
    const size_t smem = (size_t)MB * 32 * kRowE * 2 + (size_t)4 * 16 * kEpT * sizeof(float) + (size_t)4 * 16 * kRowE * 2 + (size_t)MB * 32 * kWRowE * 2;
    // whole 16-byte pieces everywhere (the tuned path) vs rows with a partly valid last piece / 2-byte aligned rows
    const bool rag = p.seqlen % 8 != 0;
    if (smem > 64 * 1024) {   // admitted per kernel and per device before the first launch there
        static PerDeviceOnce attr_once;
        const hipError_t arc = attr_once.run([&]() -> hipError_t {
            hipError_t e = hipSuccess;
#define VMS_PCB_ATTR(A_, R_)                                                                                                     \
            if (e == hipSuccess)                                                                                                \
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(&proj_conv_bwd_kernel<T, KS, A_, R_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)
            VMS_PCB_ATTR(true, true); VMS_PCB_ATTR(true, false); VMS_PCB_ATTR(false, true); VMS_PCB_ATTR(false, false);
#undef VMS_PCB_ATTR
            return e;
        });
        if (arc != hipSuccess) {
            set_error("hipFuncSetAttribute(proj_conv_bwd, %zu bytes of LDS) failed: %s", smem, hipGetErrorString(arc));
            return VMS_ERR_LAUNCH;
        }
    }
#define VMS_PCB_LAUNCH(A_, R_) hipLaunchKernelGGL((proj_conv_bwd_kernel<T, KS, A_, R_>), grid, block, smem, stream, p, tpw, pg, entry)
    if (p.dx_accumulate) { if (rag) VMS_PCB_LAUNCH(true, true); else VMS_PCB_LAUNCH(true, false); }
    else { if (rag) VMS_PCB_LAUNCH(false, true); else VMS_PCB_LAUNCH(false, false); }
#undef VMS_PCB_LAUNCH
    VMS_LAUNCH_CHECK();
    set_last_kernel(rag ? "proj_conv_bwd_ragged" : "proj_conv_bwd");
    return VMS_OK;
}

template <typename T>
static int dispatch_conv_bwd(const vms_proj_conv_bwd_params& p, hipStream_t stream, const int entry) {
    switch ((p.k + 15) / 16) {
        case 1:
        case 2: return launch_conv_bwd<T, 2>(p, stream, entry);   // k <= 32 (dt_rank + 2 d_state of the suite's d_state = 4 model: 20): one 32-deep k step
        case 3: return launch_conv_bwd<T, 3>(p, stream, entry);
        case 4: return launch_conv_bwd<T, 4>(p, stream, entry);
        case 5: return launch_conv_bwd<T, 5>(p, stream, entry);
        default: return launch_conv_bwd<T, 6>(p, stream, entry);
    }
}

}  // namespace vms

using namespace vms;

// Batch entries that follow each other without a gap in both operands (batch stride == seqlen: the blocks' channel-slowest
// tensors, x_dbl stored row-major over (batch, position)) are ONE problem of batch x seqlen positions: these products know no
// sequence boundary.  What it buys is short sequences -- TimeMamba's (1568, 8, 768): tiles of 64 positions of one batch entry are an
// eighth full there (vms_proj_wgrad 200 -> 20 us).
extern "C" int vms_proj_apply(const vms_proj_apply_params* pp, void* stream) {
    VMS_CHECK(pp != nullptr, "null parameter block");
    vms_proj_apply_params folded;
    if (pp->batch > 1 && pp->seqlen > 0 && pp->seqlen < 64 && pp->in_batch_stride == pp->seqlen && pp->out_batch_stride == pp->seqlen &&
        (int64_t)pp->batch * pp->seqlen < ((int64_t)1 << 30)) {
        folded = *pp;
        folded.seqlen = pp->batch * pp->seqlen;
        folded.batch = 1;
        pp = &folded;
    }
    const vms_proj_apply_params& p = *pp;
    VMS_CHECK(p.dtype == VMS_BF16 || p.dtype == VMS_F16, "proj_apply: 16-bit activations only (bf16 / fp16)");
    VMS_CHECK(p.batch > 0 && p.rows > 0 && p.seqlen > 0, "empty problem");
    VMS_CHECK(p.k >= 1 && p.k <= 96, "proj_apply: 1 <= k <= 96");
    VMS_CHECK(p.w && p.in && p.out, "w, in and out are required");
    VMS_CHECK(p.seqlen % 8 == 0 && p.in_batch_stride % 8 == 0 && p.in_k_stride % 8 == 0 && p.out_batch_stride % 8 == 0 &&
                  p.out_row_stride % 8 == 0 && aligned16(p.in) && aligned16(p.out),
              "proj_apply: seqlen, strides (elements) must be multiples of 8 and in / out 16-byte aligned");
    VMS_CHECK(p.in_k_stride >= 0 && p.out_row_stride >= 0 && ((int64_t)(p.k - 1) * p.in_k_stride + p.seqlen) * 2 < ((int64_t)1 << 31) &&
                  ((int64_t)(p.rows - 1) * p.out_row_stride + p.seqlen) * 2 < ((int64_t)1 << 31),
              "proj_apply: a batch entry of in and out must each span < 2 GiB (one buffer resource each)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    return p.dtype == VMS_BF16 ? dispatch_apply<bf16_t>(p, s) : dispatch_apply<f16_t>(p, s);
}

extern "C" int vms_proj_wgrad(const vms_proj_wgrad_params* pp, void* stream) {
    VMS_CHECK(pp != nullptr, "null parameter block");
    vms_proj_wgrad_params folded;      // (see vms_proj_apply)
    if (pp->batch > 1 && pp->seqlen > 0 && pp->seqlen < 64 && pp->p_batch_stride == pp->seqlen && pp->q_batch_stride == pp->seqlen &&
        (int64_t)pp->batch * pp->seqlen < ((int64_t)1 << 30)) {
        folded = *pp;
        folded.seqlen = pp->batch * pp->seqlen;
        folded.batch = 1;
        pp = &folded;
    }
    const vms_proj_wgrad_params& p = *pp;
    VMS_CHECK(p.dtype == VMS_BF16 || p.dtype == VMS_F16, "proj_wgrad: 16-bit activations only (bf16 / fp16)");
    VMS_CHECK(p.batch > 0 && p.n > 0 && p.seqlen > 0, "empty problem");
    VMS_CHECK(p.m >= 1 && p.m <= 128, "proj_wgrad: 1 <= m <= 128");
    VMS_CHECK(p.p && p.q && p.dw, "p, q and dw are required");
    VMS_CHECK(p.seqlen % 8 == 0 && p.p_batch_stride % 8 == 0 && p.p_row_stride % 8 == 0 && p.q_batch_stride % 8 == 0 &&
                  p.q_row_stride % 8 == 0 && aligned16(p.p) && aligned16(p.q),
              "proj_wgrad: seqlen, strides (elements) must be multiples of 8 and p / q 16-byte aligned");
    VMS_CHECK(p.p_row_stride >= 0 && p.q_row_stride >= 0 && ((int64_t)(p.m - 1) * p.p_row_stride + p.seqlen) * 2 < ((int64_t)1 << 31) &&
                  ((int64_t)(p.n - 1) * p.q_row_stride + p.seqlen) * 2 < ((int64_t)1 << 31),
              "proj_wgrad: a batch entry of p and q must each span < 2 GiB (one buffer resource each)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    return p.dtype == VMS_BF16 ? dispatch_wgrad<bf16_t>(p, s) : dispatch_wgrad<f16_t>(p, s);
}

extern "C" int vms_proj_kred(const vms_proj_kred_params* pp, void* stream) {
    VMS_CHECK(pp != nullptr, "null parameter block");
    const vms_proj_kred_params& p = *pp;
    VMS_CHECK(p.dtype == VMS_BF16 || p.dtype == VMS_F16, "proj_kred: 16-bit activations only (bf16 / fp16)");
    VMS_CHECK(p.batch > 0 && p.k > 0 && p.seqlen > 0, "empty problem");
    VMS_CHECK(p.m >= 1 && p.m <= 96, "proj_kred: 1 <= m <= 96");
    VMS_CHECK(p.w && p.in && p.out, "w, in and out are required");
    VMS_CHECK((p.w2 != nullptr) == (p.in2 != nullptr) && (p.in2 != nullptr) == (p.out2 != nullptr), "proj_kred: w2, in2 and out2 come together");
    VMS_CHECK(p.seqlen % 8 == 0 && p.in_batch_stride % 8 == 0 && p.in_k_stride % 8 == 0 && aligned16(p.in) && (!p.in2 || aligned16(p.in2)),
              "proj_kred: seqlen and the strides of in (elements) must be multiples of 8, in 16-byte aligned");
    // the weight moves in 16-byte pieces along its contiguous axis: (m, k) with k contiguous, or (k, m) with m contiguous
    VMS_CHECK((p.w_k_stride == 1 && p.w_row_stride % 8 == 0 && p.k % 8 == 0) || (p.w_row_stride == 1 && p.w_k_stride % 8 == 0 && p.m % 8 == 0),
              "proj_kred: w needs a unit stride along k (k and the row stride multiples of 8) or along m (m and the k stride multiples of 8)");
    VMS_CHECK(aligned16(p.w) && (!p.w2 || aligned16(p.w2)), "proj_kred: w must be 16-byte aligned");
    VMS_CHECK(!p.cast_src2 || (p.cast_src && p.in2), "proj_kred: cast_src2 comes with cast_src and the second problem");
    VMS_CHECK(!p.cast_src || (p.cast_rows > 0 && p.cast_groups > 0 && aligned16(p.cast_src) && (!p.cast_src2 || aligned16(p.cast_src2)) &&
                              p.cast_group_stride % 4 == 0 && p.cast_batch_stride % 4 == 0 && p.cast_row_stride % 4 == 0 &&
                              p.out_row_stride % 4 == 0 && p.out_batch_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(p.out) & 7) == 0 &&
                              (!p.out2 || (reinterpret_cast<uintptr_t>(p.out2) & 7) == 0)),
              "proj_kred: cast rows need cast_rows, cast_groups > 0, 16-byte aligned fp32 rows (strides multiples of 4) and 8-byte aligned rows of out");
    VMS_CHECK(((int64_t)(p.k - 1) * p.in_k_stride + p.seqlen) * 2 < ((int64_t)1 << 31) && p.in_k_stride >= 0 && p.w_row_stride >= 0 && p.w_k_stride >= 0 &&
                  ((int64_t)(p.m - 1) * p.w_row_stride + (int64_t)(p.k - 1) * p.w_k_stride + 1) * 2 < ((int64_t)1 << 31),
              "proj_kred: a batch entry of in and w must each span < 2 GiB (one buffer resource each), strides >= 0");
    hipStream_t s = static_cast<hipStream_t>(stream);
    return p.dtype == VMS_BF16 ? dispatch_kred<bf16_t>(p, s) : dispatch_kred<f16_t>(p, s);
}
extern "C" int vms_sizeof_proj_kred_params(void) { return (int)sizeof(vms_proj_kred_params); }
extern "C" int vms_conv_xproj_dual(const vms_conv_xproj_dual_params* qq, void* stream) {
    VMS_CHECK(qq != nullptr, "null parameter block");
    // gap-free batches of 8 / 16-position sequences as ONE row whose entries the kernel keeps apart (see vms_proj_conv_bwd)
    vms_conv_xproj_dual_params folded;
    int entry = 0;
    {
        const vms_conv_fwd_params& f = qq->c.f;
        if (f.batch > 1 && (f.seqlen == 8 || f.seqlen == 16) && !f.reverse && !f.reverse_from && f.x_l_stride == 1 && f.out_l_stride == 1 &&
            f.x_batch_stride == f.seqlen && f.out_batch_stride == f.seqlen && qq->c.out_b_batch_stride == f.seqlen &&
            qq->xdbl_batch_stride == f.seqlen && (int64_t)f.batch * f.seqlen < ((int64_t)1 << 30)) {
            folded = *qq;
            entry = f.seqlen;
            folded.c.f.seqlen = f.batch * f.seqlen;
            folded.c.f.batch = 1;
            qq = &folded;
        }
    }
    const vms_conv_xproj_dual_params& q = *qq;
    const vms_conv_fwd_params& p = q.c.f;
    VMS_CHECK(p.dtype == VMS_BF16 || p.dtype == VMS_F16, "conv_xproj_dual: 16-bit activations only (bf16 / fp16)");
    VMS_CHECK(p.wdtype == VMS_F32 || p.wdtype == p.dtype, "conv_xproj_dual: conv weights in fp32 or in the activations' dtype");
    VMS_CHECK(p.batch > 0 && p.dim > 0 && p.seqlen > 0, "empty problem");
    VMS_CHECK(p.width >= 2 && p.width <= 4, "causal_conv1d only supports width between 2 and 4");
    VMS_CHECK(p.silu_activation && !p.reverse && !p.reverse_from && !p.conv_state, "conv_xproj_dual: SiLU on, seqlen-contiguous layout, no reverse / reverse_from / conv_state");
    VMS_CHECK(p.x && p.weight && p.out && q.c.weight_b && q.c.out_b && q.w_x && q.w_x_b && q.x_dbl && q.x_dbl_b, "x, weight, out, weight_b, out_b, w_x, w_x_b, x_dbl, x_dbl_b are required");
    VMS_CHECK((p.bias == nullptr) == (q.c.bias_b == nullptr), "bias and bias_b come together");
    VMS_CHECK(q.m >= 1 && q.m <= 96, "conv_xproj_dual: 1 <= m <= 96");
    VMS_CHECK(p.x_l_stride == 1 && p.out_l_stride == 1, "conv_xproj_dual: unit seqlen strides");
    VMS_CHECK(p.seqlen % 8 == 0 && p.dim % 8 == 0 && p.x_batch_stride % 8 == 0 && p.x_c_stride % 8 == 0 && p.out_batch_stride % 8 == 0 && p.out_c_stride % 8 == 0 &&
                  q.c.out_b_batch_stride % 8 == 0 && q.c.out_b_c_stride % 8 == 0 && q.wx_row_stride % 8 == 0 && aligned16(p.x) && aligned16(p.out) &&
                  aligned16(q.c.out_b) && aligned16(q.w_x) && aligned16(q.w_x_b),
              "conv_xproj_dual: seqlen, dim and the strides (elements) must be multiples of 8, bases 16-byte aligned");
    {
        const int64_t lim = (int64_t)1 << 31;
        VMS_CHECK(((int64_t)(p.dim - 1) * p.x_c_stride + p.seqlen) * 2 < lim && ((int64_t)(p.dim - 1) * p.out_c_stride + p.seqlen) * 2 < lim &&
                      ((int64_t)(p.dim - 1) * q.c.out_b_c_stride + p.seqlen) * 2 < lim && p.x_c_stride >= p.seqlen && p.out_c_stride >= p.seqlen &&
                      q.c.out_b_c_stride >= p.seqlen && ((int64_t)(q.m - 1) * q.wx_row_stride + p.dim) * 2 < lim && q.wx_row_stride >= p.dim,
                  "conv_xproj_dual: a batch entry of x / out / out_b and w_x must each span < 2 GiB, rows must not overlap");
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (p.dtype == VMS_BF16) return p.wdtype == VMS_F32 ? dispatch_conv_xproj<bf16_t, float>(q, s, entry) : dispatch_conv_xproj<bf16_t, bf16_t>(q, s, entry);
    return p.wdtype == VMS_F32 ? dispatch_conv_xproj<f16_t, float>(q, s, entry) : dispatch_conv_xproj<f16_t, f16_t>(q, s, entry);
}
extern "C" int vms_sizeof_conv_xproj_dual_params(void) { return (int)sizeof(vms_conv_xproj_dual_params); }
extern "C" int vms_sizeof_proj_apply_params(void) { return (int)sizeof(vms_proj_apply_params); }
extern "C" int vms_sizeof_proj_wgrad_params(void) { return (int)sizeof(vms_proj_wgrad_params); }

extern "C" int vms_proj_conv_bwd(const vms_proj_conv_bwd_params* pp, void* stream) {
    VMS_CHECK(pp != nullptr, "null parameter block");
    // Batches of short sequences that follow each other without a gap in every operand (8 or 16 positions per entry: TimeMamba's
    // scans along time in the blocks' layouts) run as ONE row of batch x seqlen positions whose entries the kernel keeps apart
    // (`entry`): its tiles are 64 positions of one row -- an eighth full at 8 frames (2 x 480 us of a 2.5 ms block step).
    vms_proj_conv_bwd_params folded;
    int entry = 0;
    if (pp->batch > 1 && (pp->seqlen == 8 || pp->seqlen == 16) && (pp->reverse_from == 0 || pp->reverse_from >= pp->batch) &&
        pp->x_batch_stride == pp->seqlen && pp->du_batch_stride == pp->seqlen && pp->dx_batch_stride == pp->seqlen &&
        pp->dxdbl_batch_stride == pp->seqlen && (int64_t)pp->batch * pp->seqlen < ((int64_t)1 << 30)) {
        folded = *pp;
        entry = pp->seqlen;
        folded.seqlen = pp->batch * pp->seqlen;
        folded.batch = 1;
        folded.reverse_from = 0;
        pp = &folded;
    }
    const vms_proj_conv_bwd_params& p = *pp;
    VMS_CHECK(p.dtype == VMS_BF16 || p.dtype == VMS_F16, "proj_conv_bwd: 16-bit activations only (bf16 / fp16)");
    VMS_CHECK(p.wdtype == VMS_F32 || p.wdtype == VMS_F16 || p.wdtype == VMS_BF16, "conv weight dtype must be fp32/fp16/bf16");
    VMS_CHECK(p.batch > 0 && p.dim > 0 && p.seqlen > 0, "empty problem");
    VMS_CHECK(p.k >= 1 && p.k <= 96, "proj_conv_bwd: 1 <= k <= 96");
    VMS_CHECK(p.width >= 2 && p.width <= 4, "causal_conv1d only supports width between 2 and 4");
    VMS_CHECK(p.x && p.du && p.dx_dbl && p.w_x && p.conv_weight && p.dx && p.dconv_weight && p.dw_x, "x, du, dx_dbl, w_x, conv_weight, dx, dconv_weight, dw_x are required");
    VMS_CHECK(!(p.reverse && p.reverse_from), "reverse and reverse_from are exclusive");
    {
        const int64_t lim = (int64_t)1 << 31;   // a batch entry of each tensor is addressed through one buffer resource
        VMS_CHECK(p.dim * p.x_c_stride * 2 < lim && p.dim * p.du_c_stride * 2 < lim && p.dim * p.dx_c_stride * 2 < lim && p.k * p.dxdbl_k_stride * 2 < lim &&
                      p.x_c_stride >= p.seqlen && p.du_c_stride >= p.seqlen && p.dx_c_stride >= p.seqlen && p.dxdbl_k_stride >= p.seqlen,
                  "proj_conv_bwd: a batch entry (dim * channel stride) must span < 2 GiB, rows must not overlap");
    }
    // seqlen % 8 == 0: every row piece is a whole 16-byte vector and must be 16-byte aligned; otherwise (ragged rows) the vectors
    // are served at 2-byte alignment and the row's last piece moves element by element
    VMS_CHECK(p.seqlen % 8 != 0 ||
                  (p.x_batch_stride % 8 == 0 && p.x_c_stride % 8 == 0 && p.du_batch_stride % 8 == 0 && p.du_c_stride % 8 == 0 &&
                   p.dx_batch_stride % 8 == 0 && p.dx_c_stride % 8 == 0 && p.dxdbl_batch_stride % 8 == 0 && p.dxdbl_k_stride % 8 == 0 &&
                   aligned16(p.x) && aligned16(p.du) && aligned16(p.dx) && aligned16(p.dx_dbl)),
              "proj_conv_bwd: with seqlen % 8 == 0 the activation strides (elements) must be multiples of 8, bases 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    return p.dtype == VMS_BF16 ? dispatch_conv_bwd<bf16_t>(p, s, entry) : dispatch_conv_bwd<f16_t>(p, s, entry);
}
extern "C" int vms_sizeof_proj_conv_bwd_params(void) { return (int)sizeof(vms_proj_conv_bwd_params); }
