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

constexpr int kPT = 256;           // threads per workgroup (4 waves)
constexpr int kTL = 64;            // positions per tile
constexpr int kRowE = kTL + 8;     // LDS row pitch of a 16-bit tile, elements (144 B: 16 consecutive rows = 16 distinct 16-byte slots)
constexpr int kEpE = 32 + 4;       // LDS row pitch of the fp32 epilogue half tile, floats
// Both kernels are latency-bound streams (a wave has one tile of loads in flight), so what counts is resident waves: single
// LDS buffers (two workgroup barriers per tile instead of one) and <= 128 registers give 4 workgroups = 16 waves per CU.
#define VMS_PROJ_BOUNDS __launch_bounds__(kPT, 4)

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
    {
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

    // staging of the `in` tile: thread -> row (tid >> 3) + 32 pass, piece tid & 7
    s16x8 stg[NPASS];
    auto stage_load = [&](int t) __attribute__((always_inline)) {
        const int l = t * kTL + 8 * (tid & 7);
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int r = (tid >> 3) + 32 * ps;
            const bool ok = r < R && l < L && t < t_hi;
            const s16x8 v = *reinterpret_cast<const s16x8*>(in_b + (int64_t)(ok ? r : 0) * p.in_k_stride + (ok ? l : 0));
            stg[ps] = ok ? v : s16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    };
    auto stage_store = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int r = (tid >> 3) + 32 * ps;
            if (NPASS * 32 == KR || r < KR) *reinterpret_cast<lds_s16x8*>(in_lds + r * kRowE + 8 * (tid & 7)) = stg[ps];
        }
    };
    // transposing read: lane i of a 16-lane group supplies the address of row i / 4, columns 4 (i % 4) .. + 3 of a [4][16] block
    // and receives column i of its 4 rows (tools/tr_probe.hip).  Groups 0 / 1 = columns 0-15 / 16-31 of k rows 8 h .. 8 h + 3 (+ 4)
    const int i16 = lane & 15, g16 = lane >> 4;
    const lds_s16* const tb = in_lds + (8 * (g16 >> 1) + (i16 >> 2)) * kRowE + 16 * (g16 & 1) + 4 * (i16 & 3);
    // epilogue pieces of a 32 x 32 half tile: lane -> row (lane >> 2) + 16 pp, columns 8 (lane & 3) .. + 7
    const int er = lane >> 2, ec = 8 * (lane & 3);

    stage_load(t_lo);
    for (int t = t_lo; t < t_hi; ++t) {
        const int l0 = t * kTL;
        stage_store();
        __syncthreads();          // tile t is in LDS
        stage_load(t + 1);        // travels during the rest of the iteration
        vec_t<T, 8> prev[2][2];
        if (ACC) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    const int d = d0 + er + 16 * pp, l = l0 + 32 * j + ec;
                    const bool ok = d < p.rows && l < L;
                    prev[j][pp] = *reinterpret_cast<const vec_t<T, 8>*>(out_b + (int64_t)(ok ? d : 0) * p.out_row_stride + (ok ? l : 0));
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
                if (d < p.rows && l < L) *reinterpret_cast<vec_t<T, 8>*>(out_b + (int64_t)d * p.out_row_stride + l) = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// proj_wgrad
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int MB>
__global__ VMS_PROJ_BOUNDS void proj_wgrad_kernel(const vms_proj_wgrad_params p, const int tiles_per_wg) {
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

    s16x8 stp[MB], stq[4];
    auto stage_load = [&](int t) __attribute__((always_inline)) {
        const int lp = t * kTL + 8 * (tid & 7), lq = t * kTL + 8 * (lane & 7);
#pragma unroll
        for (int ps = 0; ps < MB; ++ps) {
            const int r = (tid >> 3) + 32 * ps;
            const bool ok = r < p.m && lp < L && t < t_hi;
            const s16x8 v = *reinterpret_cast<const s16x8*>(P_b + (int64_t)(ok ? r : 0) * p.p_row_stride + (ok ? lp : 0));
            stp[ps] = ok ? v : s16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int n = n0 + (lane >> 3) + 8 * ps;
            const bool ok = n < p.n && lq < L && t < t_hi;
            const s16x8 v = *reinterpret_cast<const s16x8*>(Q_b + (int64_t)(ok ? n : 0) * p.q_row_stride + (ok ? lq : 0));
            stq[ps] = ok ? v : s16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    };
    auto stage_store = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int ps = 0; ps < MB; ++ps)
            *reinterpret_cast<lds_s16x8*>(p_lds + ((tid >> 3) + 32 * ps) * kRowE + 8 * (tid & 7)) = stp[ps];
#pragma unroll
        for (int ps = 0; ps < 4; ++ps)
            *reinterpret_cast<lds_s16x8*>(q_lds + ((lane >> 3) + 8 * ps) * kRowE + 8 * (lane & 7)) = stq[ps];
    };
    f32x16 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[mb][v] = 0.f;

    const lds_s16* const pa = p_lds + c32 * kRowE + 8 * h;
    const lds_s16* const qa = q_lds + c32 * kRowE + 8 * h;
    stage_load(t_lo);
    for (int t = t_lo; t < t_hi; ++t) {
        stage_store();
        __syncthreads();          // tile t is in LDS
        stage_load(t + 1);        // travels during the matrix work
#pragma unroll
        for (int s = 0; s < kTL / 16; ++s) {
            const s16x8 bq = *reinterpret_cast<const lds_s16x8*>(qa + 16 * s);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const s16x8 ap = *reinterpret_cast<const lds_s16x8*>(pa + 32 * mb * kRowE + 16 * s);
                acc[mb] = Mfma32<T>::run(ap, bq, acc[mb]);
            }
        }
        __syncthreads();          // every wave has read tile t
    }
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
    // ~2 workgroups per CU; a workgroup ends with m x 128 atomics, so keep its range >= 16 tiles (1024 positions) when the row allows
    const int64_t want = 2 * (int64_t)device_cu_count();
    int tpw = (int)(((int64_t)n_tiles * n_blocks * p.batch + want - 1) / want);
    if (tpw < 16) tpw = 16;
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

}  // namespace vms

using namespace vms;

extern "C" int vms_proj_apply(const vms_proj_apply_params* pp, void* stream) {
    VMS_CHECK(pp != nullptr, "null parameter block");
    const vms_proj_apply_params& p = *pp;
    VMS_CHECK(p.dtype == VMS_BF16 || p.dtype == VMS_F16, "proj_apply: 16-bit activations only (bf16 / fp16)");
    VMS_CHECK(p.batch > 0 && p.rows > 0 && p.seqlen > 0, "empty problem");
    VMS_CHECK(p.k >= 1 && p.k <= 96, "proj_apply: 1 <= k <= 96");
    VMS_CHECK(p.w && p.in && p.out, "w, in and out are required");
    VMS_CHECK(p.seqlen % 8 == 0 && p.in_batch_stride % 8 == 0 && p.in_k_stride % 8 == 0 && p.out_batch_stride % 8 == 0 &&
                  p.out_row_stride % 8 == 0 && aligned16(p.in) && aligned16(p.out),
              "proj_apply: seqlen, strides (elements) must be multiples of 8 and in / out 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    return p.dtype == VMS_BF16 ? dispatch_apply<bf16_t>(p, s) : dispatch_apply<f16_t>(p, s);
}

extern "C" int vms_proj_wgrad(const vms_proj_wgrad_params* pp, void* stream) {
    VMS_CHECK(pp != nullptr, "null parameter block");
    const vms_proj_wgrad_params& p = *pp;
    VMS_CHECK(p.dtype == VMS_BF16 || p.dtype == VMS_F16, "proj_wgrad: 16-bit activations only (bf16 / fp16)");
    VMS_CHECK(p.batch > 0 && p.n > 0 && p.seqlen > 0, "empty problem");
    VMS_CHECK(p.m >= 1 && p.m <= 128, "proj_wgrad: 1 <= m <= 128");
    VMS_CHECK(p.p && p.q && p.dw, "p, q and dw are required");
    VMS_CHECK(p.seqlen % 8 == 0 && p.p_batch_stride % 8 == 0 && p.p_row_stride % 8 == 0 && p.q_batch_stride % 8 == 0 &&
                  p.q_row_stride % 8 == 0 && aligned16(p.p) && aligned16(p.q),
              "proj_wgrad: seqlen, strides (elements) must be multiples of 8 and p / q 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    return p.dtype == VMS_BF16 ? dispatch_wgrad<bf16_t>(p, s) : dispatch_wgrad<f16_t>(p, s);
}

extern "C" int vms_sizeof_proj_apply_params(void) { return (int)sizeof(vms_proj_apply_params); }
extern "C" int vms_sizeof_proj_wgrad_params(void) { return (int)sizeof(vms_proj_wgrad_params); }
