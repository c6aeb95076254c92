// selective_scan_fwd_pair.hip -- forward selective scan for gfx950 (wave64), state PAIRS in packed fp32.
//
// Same contract and the same decomposition as selective_scan_fwd_fast.hip (one wave per (batch, dim) row,
// a lane owns K consecutive elements, the lane aggregates are scanned across the wave with DPP, the 16
// running states live in one register), replacing selective_scan_fwd_kernel
// (mamba/csrc/selective_scan/selective_scan_fwd_kernel.cuh:67-303) for variable B/C, dstate 16.
//
// What is new: the 16 states are processed as 8 pairs held in 64-bit register pairs, so that every fp32
// multiply / fma of the recurrence is ONE v_pk_mul_f32 / v_pk_fma_f32 for two states.  Measured on gfx950
// (tools/microbench5.hip, profiles/r01_microbench_issue.txt): next to v_exp_f32 a scalar fp32 VALU op costs
// ~3.8 cycles of issue, a packed op 4.0 for twice the work; the transcendental itself 8.3.  Per (element,
// state pair) the kernel issues 5 packed ops + 2 v_exp_f32 (+ 4 integer ops widening bf16 B/C) where the
// unpaired kernel issued 10 scalar ops + 2 v_exp_f32 + 4.  The cross-lane scan stays per state (DPP does
// not exist for packed ops): the two states of a pair are scanned by one interleaved DPP sequence, which
// also provides the wait states a DPP read needs after a VALU write, without s_nop.
// K = 8 elements per lane (512-element chunks) keeps the pair arrays in ~120 VGPRs = 4 waves per SIMD.
#include "vms_common.cuh"

namespace vms {

constexpr int kPN = 16;  // dstate
constexpr int kPK = 8;   // elements per lane
constexpr int kPRows = 4;
typedef float f2 __attribute__((ext_vector_type(2)));

// the lane's K logical elements as raw 16-byte vectors; REV: stored right-to-left (vms_hip.h `reverse`)
template <typename T, bool REV>
struct RawP {
    static constexpr int EPV = 16 / sizeof(T);
    vec_t<T, EPV> v[kPK / EPV];
    // seqlen % K == 0: a lane's elements are all valid or all past the end; then it reads the (always
    // valid) start of the row instead and the caller neutralises it through delta = 0 -- no branch, and no
    // select that would pin the load's completion right behind its issue
    __device__ __forceinline__ void load(const T* __restrict__ base, uint32_t off, bool valid) {
        const vec_t<T, EPV>* vp = reinterpret_cast<const vec_t<T, EPV>*>(base + (valid ? off : 0u));
#pragma unroll
        for (int i = 0; i < kPK / EPV; ++i) v[i] = vp[i];
    }
    __device__ __forceinline__ float at(int i) const {
        const int e = REV ? kPK - 1 - i : i;
        return static_cast<float>(v[e / EPV][e % EPV]);
    }
};
template <typename T, bool REV>
__device__ __forceinline__ void store_p(T* __restrict__ ptr, const float (&in)[kPK]) {
    constexpr int EPV = 16 / sizeof(T);
    using V = vec_t<T, EPV>;
#pragma unroll
    for (int v = 0; v < kPK / EPV; ++v) {
        V t;
#pragma unroll
        for (int e = 0; e < EPV; ++e) t[e] = static_cast<T>(in[REV ? kPK - 1 - (v * EPV + e) : v * EPV + e]);
        reinterpret_cast<V*>(ptr)[v] = t;
    }
}

// inclusive 64-lane scan of two independent (a, x) monoids, interleaved: x += dpp(x) * a ; a *= dpp(a).
// Every DPP source was written at least 3 instructions earlier (the other monoid sits in between).
__device__ __forceinline__ void wave_scan_fused2(float& a0, float& x0, float& a1, float& x1) {
#define VMS_STEP(CTRL, RM)                                                         \
    "v_fmac_f32_dpp %0, %0, %1 " CTRL " row_mask:" RM " bank_mask:0xf\n\t"         \
    "v_fmac_f32_dpp %2, %2, %3 " CTRL " row_mask:" RM " bank_mask:0xf\n\t"         \
    "v_mul_f32_dpp %1, %1, %1 " CTRL " row_mask:" RM " bank_mask:0xf\n\t"          \
    "v_mul_f32_dpp %3, %3, %3 " CTRL " row_mask:" RM " bank_mask:0xf\n\t"
    asm volatile("s_nop 1\n\t" VMS_STEP("row_shr:1", "0xf") VMS_STEP("row_shr:2", "0xf") VMS_STEP("row_shr:4", "0xf")
                     VMS_STEP("row_shr:8", "0xf") VMS_STEP("row_bcast:15", "0xa") VMS_STEP("row_bcast:31", "0xc") "s_nop 1"
                 : "+v"(x0), "+v"(a0), "+v"(x1), "+v"(a1));
#undef VMS_STEP
}

// element i of a float array kept as register pairs, in both halves (becomes an op_sel modifier)
#define VMS_SPLAT2(arr, i) f2{arr[(i) / 2][(i) % 2], arr[(i) / 2][(i) % 2]}

#ifndef VMS_PAIR_MINWAVES
#define VMS_PAIR_MINWAVES 3
#endif
template <typename T, bool HZ, bool REV>
__global__ __launch_bounds__(kPRows* kWave, VMS_PAIR_MINWAVES) void scan_fwd_pair_kernel(const vms_scan_fwd_params p) {
    constexpr int K = kPK, N = kPN, CS = kWave * K;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // batch = blockIdx % batch: workgroups of one batch land on one XCD (blockIdx % 8) when batch == 8
    const int b = blockIdx.x % p.batch;
    const int d = (blockIdx.x / p.batch) * kPRows + wave;
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
    const T* const Bv = static_cast<const T*>(p.B) + (int64_t)b * p.B_batch_stride + (int64_t)g * p.B_group_stride;
    const T* const Cv = static_cast<const T*>(p.C) + (int64_t)b * p.C_batch_stride + (int64_t)g * p.C_group_stride;
    const int64_t xpitch = p.x_chunk_stride ? p.x_chunk_stride : 2 * N;
    float* const xck = static_cast<float*>(p.x) + ((int64_t)b * p.dim + d) * p.n_chunks * xpitch;
    const float Dd = p.D ? static_cast<const float*>(p.D)[d] : 0.f;
    const float bias = p.delta_bias ? static_cast<const float*>(p.delta_bias)[d] : 0.f;
    // lane n (< 16) keeps A[d][n] * log2(e) and the running state of recurrence n
    const float A_mine = static_cast<const float*>(p.A)[(int64_t)d * p.A_d_stride + (int64_t)(lane & 15) * p.A_dstate_stride] * kLog2e;
    float hreg = 0.f;

    RawP<T, REV> rB0a, rB0b, rC0a, rC0b, rB1a, rB1b, rC1a, rC1b;  // two named sets: explicit double buffering
    const int n_kchunks = (L + CS - 1) / CS;
    for (int c = 0; c < n_kchunks; ++c) {
        const int l0 = c * CS + lane * K;            // logical start of this lane's K elements
        const bool ok = l0 < L;
        const uint32_t pl0 = REV ? L - l0 - K : l0;  // physical start (seqlen % K == 0)
        const bool okn = l0 + CS < L;                // the same lane in the next chunk
        const uint32_t pl0n = REV ? L - l0 - CS - K : l0 + CS;
        if (c == 0) {
            rB0a.load(Bv, pl0, ok);
            rB0b.load(Bv + p.B_dstate_stride, pl0, ok);
            rC0a.load(Cv, pl0, ok);
            rC0b.load(Cv + p.C_dstate_stride, pl0, ok);
        }
        f2 dl2[K / 2], du2[K / 2], y2[K];
        float sdl = 0.f;
        {
            RawP<T, REV> t0, t1;
            t0.load(u_b, o_u + pl0, ok);
            t1.load(dt_b, o_dt + pl0, ok);
#pragma unroll
            for (int i = 0; i < K; ++i) {
                float t = t1.at(i) + bias;
                if (p.delta_softplus) t = softplusf_(t);
                t = ok ? t : 0.f;  // past the end: delta = 0 -> a = 1, b = 0 (identity)
                const float uv = t0.at(i);
                dl2[i / 2][i % 2] = t;
                du2[i / 2][i % 2] = t * uv;
                y2[i] = f2{Dd * uv, 0.f};
                sdl += t;
            }
        }
        auto do_pair = [&](const int q, const RawP<T, REV>& cBa, const RawP<T, REV>& cBb, const RawP<T, REV>& cCa,
                           const RawP<T, REV>& cCb, RawP<T, REV>& nBa, RawP<T, REV>& nBb, RawP<T, REV>& nCa,
                           RawP<T, REV>& nCb) __attribute__((always_inline)) {
            {   // B / C of the next pair -- after the last pair: pair 0 of the next chunk
                const int qn = (q + 1) & (N / 2 - 1);
                const bool wrap = q + 1 == N / 2;
                const T* const Bn = Bv + (int64_t)(2 * qn) * p.B_dstate_stride;
                const T* const Cn = Cv + (int64_t)(2 * qn) * p.C_dstate_stride;
                const uint32_t po = wrap ? pl0n : pl0;
                const bool pok = wrap ? okn : ok;
                nBa.load(Bn, po, pok);
                nBb.load(Bn + p.B_dstate_stride, po, pok);
                nCa.load(Cn, po, pok);
                nCb.load(Cn + p.C_dstate_stride, po, pok);
            }
            const f2 An2 = f2{readlane_f(A_mine, 2 * q), readlane_f(A_mine, 2 * q + 1)};
            const f2 hin2 = f2{readlane_f(hreg, 2 * q), readlane_f(hreg, 2 * q + 1)};
            f2 a2[K], bx2[K];
            f2 px2 = f2{0.f, 0.f};
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const f2 t = VMS_SPLAT2(dl2, i) * An2;
                a2[i] = f2{fast_exp2(t.x), fast_exp2(t.y)};
                bx2[i] = VMS_SPLAT2(du2, i) * f2{cBa.at(i), cBb.at(i)};
                px2 = __builtin_elementwise_fma(a2[i], px2, bx2[i]);
            }
            const f2 ts = f2{sdl, sdl} * An2;
            f2 pa2 = f2{fast_exp2(ts.x), fast_exp2(ts.y)};  // product of the lane's K a_i
            {
                float a0 = pa2.x, x0 = px2.x, a1 = pa2.y, x1 = px2.y;
                wave_scan_fused2(a0, x0, a1, x1);
                pa2 = f2{a0, a1};
                px2 = f2{x0, x1};
            }
            // exclusive prefix of this lane, seeded with the state carried from earlier chunks
            const f2 ea2 = f2{dpp_mov<DPP_WAVE_SHR1, 0xf>(1.f, pa2.x), dpp_mov<DPP_WAVE_SHR1, 0xf>(1.f, pa2.y)};
            const f2 ex2 = f2{dpp_mov<DPP_WAVE_SHR1, 0xf>(0.f, px2.x), dpp_mov<DPP_WAVE_SHR1, 0xf>(0.f, px2.y)};
            f2 xs2 = __builtin_elementwise_fma(ea2, hin2, ex2);
            const f2 hend2 = __builtin_elementwise_fma(pa2, hin2, px2);  // state after this lane's last element
            if (p.x_has_sub && ((lane + 1) * K) % 128 == 0) {  // 128-element sub-checkpoints for the backward kernel
                const int i128 = (c * CS + (lane + 1) * K) / 128 - 1;
                float* dst = xck + (int64_t)(i128 >> 4) * xpitch + 2 * N + (i128 & 15) * N + 2 * q;
                dst[0] = hend2.x;
                dst[1] = hend2.y;
            }
            const float hout0 = readlane_f(hend2.x, 63), hout1 = readlane_f(hend2.y, 63);
            if (lane == 2 * q) hreg = hout0;
            if (lane == 2 * q + 1) hreg = hout1;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                xs2 = __builtin_elementwise_fma(a2[i], xs2, bx2[i]);
                y2[i] = __builtin_elementwise_fma(f2{cCa.at(i), cCb.at(i)}, xs2, y2[i]);
            }
        };
#pragma unroll 1
        for (int q = 0; q < N / 2; q += 2) {
            do_pair(q, rB0a, rB0b, rC0a, rC0b, rB1a, rB1b, rC1a, rC1b);
            do_pair(q + 1, rB1a, rB1b, rC1a, rC1b, rB0a, rB0b, rC0a, rC0b);
        }
        float y[K];
#pragma unroll
        for (int i = 0; i < K; ++i) y[i] = y2[i].x + y2[i].y;
        if (ok) store_p<T, REV>(out_b + (o_out + pl0), y);
        if (HZ) {
            RawP<T, REV> tz;
            tz.load(z_b, o_z + pl0, ok);
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const float zv = tz.at(i);
                y[i] *= zv * sigmoidf_(zv);
            }
            if (ok) store_p<T, REV>(outz_b + (o_oz + pl0), y);
        }
        // reference-shaped checkpoints every 1024 elements (vms_hip.h): even slot = state after the
        // first 1024 elements of a 2048-chunk, odd slot = state after the chunk (or the sequence)
        const bool last = c == n_kchunks - 1;
        const int pos = (c + 1) * CS;
        if (lane < N && (last || pos % 1024 == 0)) {
            const int blk = last ? (L - 1) / 2048 : (pos - 1) / 2048;
            const int r = (last ? L : pos) - blk * 2048;
            float* xb = xck + (int64_t)blk * xpitch;
            if (r <= 1024) xb[2 * lane] = hreg;
            if (r == 2048 || last) xb[2 * lane + 1] = hreg;
        }
    }
}

bool scan_fwd_pair_eligible(const vms_scan_fwd_params& p, bool vec) {
    if (!vec || !p.is_variable_B || !p.is_variable_C || p.dstate != kPN) return false;
    if (p.seqlen % kPK != 0) return false;
    const int64_t lim = (int64_t)1 << 31;
    auto span = [&](int64_t bs, int64_t ds) { return (p.batch - 1) * bs + (p.dim - 1) * ds + p.seqlen; };
    if (span(p.u_batch_stride, p.u_d_stride) >= lim || span(p.delta_batch_stride, p.delta_d_stride) >= lim ||
        span(p.out_batch_stride, p.out_d_stride) >= lim || span(p.z_batch_stride, p.z_d_stride) >= lim ||
        span(p.out_z_batch_stride, p.out_z_d_stride) >= lim || (int64_t)(p.dstate - 1) * p.B_dstate_stride + p.seqlen >= lim ||
        (int64_t)(p.dstate - 1) * p.C_dstate_stride + p.seqlen >= lim)
        return false;
    return true;
}

template <typename T>
static int launch_pair(const vms_scan_fwd_params& p, hipStream_t stream) {
    const int tiles = (p.dim + kPRows - 1) / kPRows;
    dim3 grid(p.batch * tiles), block(kPRows * kWave);
#define VMS_L(Z_, R_) hipLaunchKernelGGL((scan_fwd_pair_kernel<T, Z_, R_>), grid, block, 0, stream, p)
    if (p.reverse) { if (p.z) VMS_L(true, true); else VMS_L(false, true); }
    else { if (p.z) VMS_L(true, false); else VMS_L(false, false); }
#undef VMS_L
    VMS_LAUNCH_CHECK();
    return VMS_OK;
}

int launch_scan_fwd_pair(const vms_scan_fwd_params& p, hipStream_t stream) {
    switch (p.dtype) {
        case VMS_BF16: return launch_pair<bf16_t>(p, stream);
        case VMS_F16: return launch_pair<f16_t>(p, stream);
        default: return launch_pair<float>(p, stream);
    }
}

}  // namespace vms
