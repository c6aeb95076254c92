// selective_scan_fwd_fast.hip -- the fast forward selective scan for gfx950 (wave64).
//
// Same math as selective_scan_fwd.hip (the generic kernel); taken when B and C are input
// dependent, dstate == 16, I/O is 16-byte aligned and seqlen % 16 == 0.  Replaces
// selective_scan_fwd_kernel (mamba/csrc/selective_scan/selective_scan_fwd_kernel.cuh:67-303).
//
// What differs from the generic kernel (DESIGN.md "scan forward, fast path"):
//   * still one wave per (batch, dim) row, 64 lanes x 16 elements per 1024-element chunk -- 8192
//     rows at the headline size give 8 waves per SIMD to draw from, which this chip needs (one wave
//     issues a VALU instruction only every ~8.5 cycles, tools/microbench/microbench.hip);
//   * the 64-lane scan of the lane aggregates is 6 steps of DPP-fused VOP2 pairs
//     (v_fmac_f32_dpp / v_mul_f32_dpp) in one asm block: 12 instructions instead of ~36;
//   * the lane aggregate's "a" component is exp2(A * sum(delta)) (1 mul + 1 exp) instead of a
//     16-term product;
//   * the running state of the 16 recurrences lives in one register (lane n keeps state n) and is
//     read with v_readlane -- no LDS at all in this kernel;
//   * B / C of the next state are requested (raw 16-byte vectors, two named register sets) while the
//     current state computes, and are widened to fp32 at the point of use;
//   * loads are branch-free (seqlen % 16 == 0: a lane's 16 elements are all in range or all out),
//     row addressing is a uniform base + one 32-bit offset.
#include "vms_common.h"

namespace vms {

constexpr int kFN = 16;    // dstate
#ifndef VMS_FWD_K
#define VMS_FWD_K 16
#endif
constexpr int kFK = VMS_FWD_K;  // elements per lane (4, 8 or 16)
constexpr int kFRows = 4;  // waves (rows) per workgroup

// REV: the lane's K logical elements are stored right-to-left (see vms_hip.h `reverse`): element i of
// the logical run is vector element K-1-i of the physical run
template <typename T, int K, bool REV>
struct RawVecF {
    static constexpr int EPV = (16 / sizeof(T)) < K ? (16 / sizeof(T)) : K;  // 16-byte vectors, or 8-byte when K is small
    vec_t<T, EPV> v[K / EPV];
    __device__ __forceinline__ void load(const T* __restrict__ base, uint32_t off, bool valid) {
        const vec_t<T, EPV>* vp = reinterpret_cast<const vec_t<T, EPV>*>(base + (valid ? off : 0u));
#pragma unroll
        for (int i = 0; i < K / EPV; ++i) {
            v[i] = vp[i];
            if (!valid) v[i] = vec_t<T, EPV>{};
        }
    }
    __device__ __forceinline__ float at(int i) const {
        const int e = REV ? K - 1 - i : i;
        return static_cast<float>(v[e / EPV][e % EPV]);
    }
};

template <typename T, int K, bool REV>
__device__ __forceinline__ void store_vec(T* __restrict__ ptr, const float (&in)[K]) {
    constexpr int EPV = (16 / sizeof(T)) < K ? (16 / sizeof(T)) : K;
    using V = vec_t<T, EPV>;
#pragma unroll
    for (int v = 0; v < K / EPV; ++v) {
        V t;
#pragma unroll
        for (int e = 0; e < EPV; ++e) t[e] = static_cast<T>(in[REV ? K - 1 - (v * EPV + e) : v * EPV + e]);
        reinterpret_cast<V*>(ptr)[v] = t;
    }
}

// inclusive scan of the monoid (a, x) over the 64 lanes: 4 in-row steps + 2 cross-row broadcasts.
// x += dpp(x) * a ; a *= dpp(a); lanes without a DPP source (or masked rows) are not written.
__device__ __forceinline__ void wave_scan_fused(float& a, float& x) {
#define VMS_STEP(CTRL, RM)                                                         \
    "v_fmac_f32_dpp %0, %0, %1 " CTRL " row_mask:" RM " bank_mask:0xf\n\t"         \
    "v_mul_f32_dpp %1, %1, %1 " CTRL " row_mask:" RM " bank_mask:0xf\n\t"          \
    "s_nop 1\n\t"
    asm volatile("s_nop 1\n\t" VMS_STEP("row_shr:1", "0xf") VMS_STEP("row_shr:2", "0xf") VMS_STEP("row_shr:4", "0xf")
                     VMS_STEP("row_shr:8", "0xf") VMS_STEP("row_bcast:15", "0xa") VMS_STEP("row_bcast:31", "0xc")
                 : "+v"(x), "+v"(a));
#undef VMS_STEP
}

#ifndef VMS_FWD_MINWAVES
#define VMS_FWD_MINWAVES 3  // <= 168 VGPRs: 3 waves per SIMD (measured best of 2 / 3 / 4)
#endif
template <typename T, bool HZ, bool REV>
__global__ __launch_bounds__(kFRows* kWave, VMS_FWD_MINWAVES) void scan_fwd_fast_kernel(const vms_scan_fwd_params p) {
    constexpr int K = kFK, N = kFN, CS = kWave * K;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // batch = blockIdx % batch: workgroups of one batch land on one XCD (blockIdx % 8) when batch == 8
    const int b = blockIdx.x % p.batch;
    const int d = (blockIdx.x / p.batch) * kFRows + wave;
    if (d >= p.dim) return;  // no barriers in this kernel
    const int g = d / (p.dim / p.n_groups);
    const int L = p.seqlen;

    const T* const u_b = static_cast<const T*>(p.u);
    const T* const dt_b = static_cast<const T*>(p.delta);
    T* const out_b = static_cast<T*>(p.out);
    const T* const z_b = static_cast<const T*>(p.z);
    T* const outz_b = static_cast<T*>(p.out_z);
    const uint32_t o_u = static_cast<uint32_t>((int64_t)b * p.u_batch_stride + (int64_t)d * p.u_d_stride);
    const uint32_t o_dt = static_cast<uint32_t>((int64_t)b * p.delta_batch_stride + (int64_t)d * p.delta_d_stride);
    const uint32_t o_out = static_cast<uint32_t>((int64_t)b * p.out_batch_stride + (int64_t)d * p.out_d_stride);
    const uint32_t o_z = HZ ? static_cast<uint32_t>((int64_t)b * p.z_batch_stride + (int64_t)d * p.z_d_stride) : 0u;
    const uint32_t o_oz = HZ ? static_cast<uint32_t>((int64_t)b * p.out_z_batch_stride + (int64_t)d * p.out_z_d_stride) : 0u;
    const T* Bv = static_cast<const T*>(p.B) + (int64_t)b * p.B_batch_stride + (int64_t)g * p.B_group_stride;
    const T* Cv = static_cast<const T*>(p.C) + (int64_t)b * p.C_batch_stride + (int64_t)g * p.C_group_stride;
    const int64_t xpitch = p.x_chunk_stride ? p.x_chunk_stride : 2 * N;
    float* xck = static_cast<float*>(p.x) + ((int64_t)b * p.dim + d) * p.n_chunks * xpitch;
    const float Dd = p.D ? static_cast<const float*>(p.D)[d] : 0.f;
    const float bias = p.delta_bias ? static_cast<const float*>(p.delta_bias)[d] : 0.f;
    // lane n (< 16) keeps A[d][n] * log2(e) and the running state of recurrence n
    const float A_mine = static_cast<const float*>(p.A)[(int64_t)d * p.A_d_stride + (int64_t)(lane & 15) * p.A_dstate_stride] * kLog2e;
    float hreg = 0.f;

    const int n_kchunks = (L + CS - 1) / CS;
    for (int c = 0; c < n_kchunks; ++c) {
        const int l0 = c * CS + lane * K;        // logical start of this lane's K elements
        const bool ok = l0 < L;
        const uint32_t pl0 = REV ? L - l0 - K : l0;  // physical start (seqlen % K == 0)
        const T* const Bc = Bv;
        const T* const Cc = Cv;
        const uint32_t jo = pl0;
        RawVecF<T, K, REV> rB0, rC0, rB1, rC1;
        rB0.load(Bc, jo, ok);
        rC0.load(Cc, jo, ok);
        float dl[K], du[K], y[K];
        float sdl = 0.f;
        {
            RawVecF<T, K, REV> t0, t1;
            t0.load(u_b, o_u + pl0, ok);
            t1.load(dt_b, o_dt + pl0, ok);
#pragma unroll
            for (int i = 0; i < K; ++i) {
                float t = t1.at(i) + bias;
                if (p.delta_softplus) t = softplusf_(t);
                dl[i] = ok ? t : 0.f;  // past the end: delta = 0 -> a = 1, b = 0 (identity)
                const float uv = t0.at(i);
                du[i] = dl[i] * uv;
                y[i] = Dd * uv;
                sdl += dl[i];
            }
        }
        auto do_state = [&](const int n, const RawVecF<T, K, REV>& cB, const RawVecF<T, K, REV>& cC,
                            RawVecF<T, K, REV>& nB, RawVecF<T, K, REV>& nC) __attribute__((always_inline)) {
            if (n + 1 < N) {
                nB.load(Bc + (int64_t)(n + 1) * p.B_dstate_stride, jo, ok);
                nC.load(Cc + (int64_t)(n + 1) * p.C_dstate_stride, jo, ok);
            }
            const float An = readlane_f(A_mine, n);
            const float hin = readlane_f(hreg, n);
            float a[K], bx[K];
            float px = 0.f;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                a[i] = fast_exp2(dl[i] * An);
                bx[i] = du[i] * cB.at(i);
                px = fmaf(a[i], px, bx[i]);
            }
            float pa = fast_exp2(sdl * An);  // product of the lane's 16 a_i
            wave_scan_fused(pa, px);
            // exclusive prefix of this lane, seeded with the state carried from earlier chunks
            const float ea = dpp_mov<DPP_WAVE_SHR1, 0xf>(1.f, pa);
            const float ex = dpp_mov<DPP_WAVE_SHR1, 0xf>(0.f, px);
            float xs = fmaf(ea, hin, ex);
            const float hend = fmaf(pa, hin, px);  // state after this lane's last element
            if (p.x_has_sub && ((lane + 1) * K) % 128 == 0) {  // 128-element sub-checkpoints for the backward kernel
                const int i128 = (c * CS + (lane + 1) * K) / 128 - 1;
                xck[(int64_t)(i128 >> 4) * xpitch + 2 * N + (i128 & 15) * N + n] = hend;
            }
            const float hout = readlane_f(hend, 63);
            if (lane == n) hreg = hout;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                xs = fmaf(a[i], xs, bx[i]);
                y[i] = fmaf(cC.at(i), xs, y[i]);
            }
        };
#pragma unroll 1
        for (int n = 0; n < N; n += 2) {
            do_state(n, rB0, rC0, rB1, rC1);
            do_state(n + 1, rB1, rC1, rB0, rC0);
        }
        if (ok) store_vec<T, K, REV>(out_b + (o_out + pl0), y);
        if (HZ) {
            RawVecF<T, K, REV> tz, told;
            tz.load(z_b, o_z + pl0, ok);
            if (p.out_z_accumulate) told.load(outz_b, o_oz + pl0, ok);  // out_z += (vms_hip.h)
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const float zv = tz.at(i);
                y[i] *= zv * sigmoidf_(zv);
            }
            if (p.out_z_accumulate) {
#pragma unroll
                for (int i = 0; i < K; ++i) y[i] += told.at(i);
            }
            if (ok) store_vec<T, K, REV>(outz_b + (o_oz + pl0), y);
        }
        // reference-shaped checkpoints every 1024 elements (vms_hip.h): even slot = state after the
        // first 1024 elements of a 2048-chunk, odd slot = state after the chunk (or the sequence)
        const bool last = c == n_kchunks - 1;
        const int pos = (c + 1) * CS;
        const int blk = last ? (L - 1) / 2048 : (pos - 1) / 2048;
        const int r = (last ? L : pos) - blk * 2048;
        if (lane < N && (last || pos % 1024 == 0)) {
            float* xb = xck + (int64_t)blk * xpitch;
            if (r <= 1024) xb[2 * lane] = hreg;
            if (r == 2048 || last) xb[2 * lane + 1] = hreg;
        }
    }
}

bool scan_fwd_fast_eligible(const vms_scan_fwd_params& p, bool vec) {
    if (!vec || !p.is_variable_B || !p.is_variable_C || p.dstate != kFN || p.x_has_sub == 3) return false;
    if (p.seqlen % kFK != 0) return false;
    const int64_t lim = (int64_t)1 << 31;
    auto span = [&](int64_t bs, int64_t ds) { return (p.batch - 1) * bs + (p.dim - 1) * ds + p.seqlen; };
    if (span(p.u_batch_stride, p.u_d_stride) >= lim || span(p.delta_batch_stride, p.delta_d_stride) >= lim ||
        span(p.out_batch_stride, p.out_d_stride) >= lim || span(p.z_batch_stride, p.z_d_stride) >= lim ||
        span(p.out_z_batch_stride, p.out_z_d_stride) >= lim)
        return false;
    return true;
}

template <typename T>
static int launch_fast(const vms_scan_fwd_params& p, hipStream_t stream) {
    const int tiles = (p.dim + kFRows - 1) / kFRows;
    dim3 grid(p.batch * tiles), block(kFRows * kWave);
#define VMS_L(Z_, R_) hipLaunchKernelGGL((scan_fwd_fast_kernel<T, Z_, R_>), grid, block, 0, stream, p)
    if (p.reverse) { if (p.z) VMS_L(true, true); else VMS_L(false, true); }
    else { if (p.z) VMS_L(true, false); else VMS_L(false, false); }
#undef VMS_L
    VMS_LAUNCH_CHECK();
    return VMS_OK;
}

int launch_scan_fwd_fast(const vms_scan_fwd_params& p, hipStream_t stream) {
    switch (p.dtype) {
        case VMS_BF16: return launch_fast<bf16_t>(p, stream);
        case VMS_F16: return launch_fast<f16_t>(p, stream);
        default: return launch_fast<float>(p, stream);
    }
}

}  // namespace vms
