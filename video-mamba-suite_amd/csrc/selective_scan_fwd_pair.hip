// selective_scan_fwd_pair.hip -- forward selective scan for gfx950 (wave64), ELEMENT PAIRS in packed fp32.
//
// Same contract and the same decomposition as selective_scan_fwd_fast.hip (one wave per (batch, dim) row,
// a lane owns K consecutive elements, the lane aggregates are scanned across the wave with DPP, the 16
// running states live in one register), replacing selective_scan_fwd_kernel
// (mamba/csrc/selective_scan/selective_scan_fwd_kernel.cuh:67-303) for variable B/C, dstate 16.
//
// What is new: everything that is independent across the 16 elements of a lane (the exp arguments,
// b = delta u B, the contraction y += C x) runs on PAIRS of consecutive elements with v_pk_mul_f32 /
// v_pk_fma_f32; the two recurrences along the elements stay scalar fma chains.  Measured on gfx950
// (tools/microbench/microbench5.hip, profiles/r01_microbench_issue.txt): next to v_exp_f32 a scalar fp32 VALU op costs
// ~3.8 cycles of issue, a packed op 4.0 for twice the work.  (Pairing the STATES instead -- also tried --
// packs the chains too but doubles every per-element array: it has to drop to 8 elements per lane, which
// doubles the cross-lane scan cost per element, and came out slower.)  Loads carry no select (a lane past the
// end reads the row start and is neutralised through delta = 0), so that the prefetch of the next state's
// B / C -- after the last state: of the next chunk's first state -- really stays in flight.
#include "vms_common.h"
#include <type_traits>

#ifndef VMS_FWD_LD_NT
#define VMS_FWD_LD_NT 0
#endif
#ifndef VMS_FWD_ST_NT
#define VMS_FWD_ST_NT 0   /* 1 (A/B builds): out / out_z leave as streaming (nt) stores */
#endif
namespace vms {

constexpr int kPN = 16;  // dstate
constexpr int kPK = 16;  // elements per lane
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
    // row data this launch touches once (u, delta, z): VMS_FWD_LD_NT=1 (A/B builds) loads it as streaming data
    __device__ __forceinline__ void load_row(const T* __restrict__ base, uint32_t off, bool valid) {
        const vec_t<T, EPV>* vp = reinterpret_cast<const vec_t<T, EPV>*>(base + (valid ? off : 0u));
#pragma unroll
        for (int i = 0; i < kPK / EPV; ++i) v[i] = VMS_FWD_LD_NT ? __builtin_nontemporal_load(&vp[i]) : vp[i];
    }
    // signed offset: a partly valid vector of a padded B / C row may start before the row (REV) -- vms_hip.h bc_pad.
    // `safe`: where a lane beyond the row reads instead -- the piece of the row's FIRST lane, which the padding covers: offset 0
    // left-to-right, seqlen - K right-to-left.  (Until round 5 such lanes read [0, K) of the row in both directions: K - seqlen
    // elements past the end of a front-padded row shorter than K -- harmless zeros of the next row, except behind the tensor's
    // last row: a memory fault when the padded copy ended a mapped segment; found by tools/fuzz_modules.py, tools/guard_probe.py.)
    __device__ __forceinline__ void load_s(const T* __restrict__ base, int32_t off, bool valid, int32_t safe) {
        const vec_t<T, EPV>* vp = reinterpret_cast<const vec_t<T, EPV>*>(base + (valid ? off : safe));
#pragma unroll
        for (int i = 0; i < kPK / EPV; ++i) v[i] = vp[i];
    }
    // ragged rows: the nv (< K) valid logical elements [l0, l0 + nv) of the row one by one, zeros behind them
    __device__ __forceinline__ void load_partial(const T* __restrict__ row, int l0, int L, int nv) {
#pragma unroll
        for (int i = 0; i < kPK; ++i) {
            const int e = REV ? kPK - 1 - i : i;
            v[e / EPV][e % EPV] = i < nv ? row[REV ? L - 1 - (l0 + i) : l0 + i] : static_cast<T>(0.f);
        }
    }
    __device__ __forceinline__ float at(int i) const {
        const int e = REV ? kPK - 1 - i : i;
        return static_cast<float>(v[e / EPV][e % EPV]);
    }
};
template <typename T, bool REV>
__device__ __forceinline__ void store_partial_p(T* __restrict__ row, int l0, int L, int nv, const float (&in)[kPK]) {
#pragma unroll
    for (int i = 0; i < kPK; ++i)
        if (i < nv) row[REV ? L - 1 - (l0 + i) : l0 + i] = static_cast<T>(in[i]);
}
template <typename T, bool REV>
__device__ __forceinline__ void store_p(T* __restrict__ ptr, const float (&in)[kPK]) {
    constexpr int EPV = 16 / sizeof(T);
    using V = vec_t<T, EPV>;
#pragma unroll
    for (int v = 0; v < kPK / EPV; ++v) {
        V t;
#pragma unroll
        for (int e = 0; e < EPV; ++e) t[e] = static_cast<T>(in[REV ? kPK - 1 - (v * EPV + e) : v * EPV + e]);
        if (VMS_FWD_ST_NT) __builtin_nontemporal_store(t, reinterpret_cast<V*>(ptr) + v);
        else reinterpret_cast<V*>(ptr)[v] = t;
    }
}

// inclusive scan of the monoid (a, x) over the 64 lanes: 4 in-row steps + 2 cross-row broadcasts.
// x += dpp(x) * a ; a *= dpp(a); lanes without a DPP source (or masked rows) are not written.
__device__ __forceinline__ void wave_scan_fused_p(float& a, float& x) {
#define VMS_STEP(CTRL, RM)                                                         \
    "v_fmac_f32_dpp %0, %0, %1 " CTRL " row_mask:" RM " bank_mask:0xf\n\t"         \
    "v_mul_f32_dpp %1, %1, %1 " CTRL " row_mask:" RM " bank_mask:0xf\n\t"          \
    "s_nop 1\n\t"
    asm volatile("s_nop 1\n\t" VMS_STEP("row_shr:1", "0xf") VMS_STEP("row_shr:2", "0xf") VMS_STEP("row_shr:4", "0xf")
                     VMS_STEP("row_shr:8", "0xf") VMS_STEP("row_bcast:15", "0xa") VMS_STEP("row_bcast:31", "0xc")
                 : "+v"(x), "+v"(a));
#undef VMS_STEP
}
// the same over each 16-lane DPP row separately (4 in-row steps): four independent scans per wave
__device__ __forceinline__ void row_scan_fused_p(float& a, float& x) {
#define VMS_STEP(CTRL)                                                            \
    "v_fmac_f32_dpp %0, %0, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\t"          \
    "v_mul_f32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\t"           \
    "s_nop 1\n\t"
    asm volatile("s_nop 1\n\t" VMS_STEP("row_shr:1") VMS_STEP("row_shr:2") VMS_STEP("row_shr:4") VMS_STEP("row_shr:8") : "+v"(x), "+v"(a));
#undef VMS_STEP
}
__device__ __forceinline__ f2 pk_fma_p(f2 a, f2 b, f2 c) {
    f2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
#define VMS_ELP(arr, i) arr[(i) / 2][(i) % 2]

// softplusf_ (vms_common.h) on a pair of elements: the same operations, the non-transcendental ones as packed fp32 (next to
// transcendentals a scalar VALU op costs 3.8 cycles of the SIMD, a packed one 4.0 for two lane-ops: DESIGN.md 4.0)
__device__ __forceinline__ f2 softplus2_p(f2 x) {
    const f2 arg = x * f2{kLog2e, kLog2e};
    const f2 t = f2{fast_exp2(arg.x), fast_exp2(arg.y)};
    const f2 one = f2{1.f, 1.f};
    const f2 w = one + t;
    const f2 d = t - (w - one);
    const f2 rw = f2{fast_rcp(w.x), fast_rcp(w.y)};
    const f2 lg = f2{__builtin_amdgcn_logf(w.x), __builtin_amdgcn_logf(w.y)} * f2{0.6931471805599453f, 0.6931471805599453f};
    f2 r = pk_fma_p(d, rw, lg);
    r.x = x.x <= 20.f ? r.x : x.x;
    r.y = x.y <= 20.f ? r.y : x.y;
    return r;
}
// z * sigmoid(z) on a pair
__device__ __forceinline__ f2 silu2_p(f2 z) {
    const f2 arg = z * f2{-kLog2e, -kLog2e};
    const f2 w = f2{1.f, 1.f} + f2{fast_exp2(arg.x), fast_exp2(arg.y)};
    return z * f2{fast_rcp(w.x), fast_rcp(w.y)};
}


#ifndef VMS_PAIR_MINWAVES
#define VMS_PAIR_MINWAVES 3
#endif
// RAG: seqlen is not a multiple of K.  The last valid lane of a row then owns nv < K elements: it reads / writes
// its activations one by one (once per row), every lane masks per element, and B / C rows are read through the
// padding the caller guarantees behind their logical end (vms_hip.h bc_pad).  RAG = false is the tuned path.
template <typename T, bool HZ, bool REV, bool RAG>
__global__ __launch_bounds__(kPRows* kWave, VMS_PAIR_MINWAVES) void scan_fwd_pair_kernel(const vms_scan_fwd_params p,
                                                                                          const int n_seg,
                                                                                          const float2* __restrict__ seg_carry) {
    constexpr int K = kPK, N = kPN, CS = kWave * K;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // n_seg > 1 (few rows, long sequences): the grid repeats n_seg times, copy `seg` walks the chunks [c_lo, c_hi) of
    // every row from the state scan_fwd_carry_kernel's (P, q) pairs give for the chunks before them
    const int wg_per_seg = gridDim.x / n_seg;
    const int seg = blockIdx.x / wg_per_seg, wg = blockIdx.x - seg * wg_per_seg;
    // batch = wg % batch: workgroups of one batch land on one XCD (blockIdx % 8) when batch == 8
    const int b = wg % p.batch;
    const int d = (wg / p.batch) * kPRows + wave;
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

    RawP<T, REV> rB0, rC0, rB1, rC1;  // two named sets: explicit double buffering
    const int n_kchunks = (L + CS - 1) / CS;
    const int cps = (n_kchunks + n_seg - 1) / n_seg;
    const int c_lo = seg * cps, c_hi = (c_lo + cps < n_kchunks) ? c_lo + cps : n_kchunks;
    if (seg > 0 && lane < N) {  // state entering this range: x = P x + q over the ranges to its left
        const float2* cp = seg_carry + (((int64_t)b * p.dim + d) * n_seg) * N + lane;
        for (int s2 = 0; s2 < seg; ++s2) {
            const float2 pq = cp[(int64_t)s2 * N];
            hreg = fmaf(pq.x, hreg, pq.y);
        }
    }
    for (int c = c_lo; c < c_hi; ++c) {
        const int l0 = c * CS + lane * K;            // logical start of this lane's K elements
        const bool ok = l0 < L;
        const int nv = RAG ? (L - l0 >= K ? K : (L - l0 > 0 ? L - l0 : 0)) : (ok ? K : 0);  // valid elements of the lane
        const bool full = !RAG || nv == K;
        const uint32_t pl0 = REV ? L - l0 - K : l0;  // physical start (as int32: negative in a partly valid REV lane)
        const bool okn = l0 + CS < L;                // the same lane in the next chunk
        const uint32_t pl0n = REV ? L - l0 - CS - K : l0 + CS;
        if (c == c_lo) {
            if (RAG) {
                rB0.load_s(Bv, (int32_t)pl0, ok, REV ? L - K : 0);
                rC0.load_s(Cv, (int32_t)pl0, ok, REV ? L - K : 0);
            } else {
                rB0.load(Bv, pl0, ok);
                rC0.load(Cv, pl0, ok);
            }
        }
        f2 dl2[K / 2], du2[K / 2], y2[K / 2];
        float sdl = 0.f;
        {
            RawP<T, REV> t0, t1;
            if (full) {
                t0.load(u_b, o_u + pl0, ok);
                t1.load(dt_b, o_dt + pl0, ok);
            } else {
                t0.load_partial(u_b + o_u, l0, L, nv);
                t1.load_partial(dt_b + o_dt, l0, L, nv);
            }
#pragma unroll
            for (int i = 0; i < K; ++i) {
                float t = t1.at(i) + bias;
                if (p.delta_softplus) t = softplusf_(t);
                t = (RAG ? i < nv : ok) ? t : 0.f;  // past the end: delta = 0 -> a = 1, b = 0 (identity)
                const float uv = t0.at(i);
                dl2[i / 2][i % 2] = t;
                du2[i / 2][i % 2] = t * uv;
                y2[i / 2][i % 2] = Dd * uv;
                sdl += t;
            }
        }
        auto do_state = [&](const int n, const RawP<T, REV>& cB, const RawP<T, REV>& cC, RawP<T, REV>& nB,
                            RawP<T, REV>& nC) __attribute__((always_inline)) {
            {   // B / C of the next state -- after the last state: state 0 of the next chunk
                const int nn = (n + 1) & (N - 1);
                const bool wrap = n + 1 == N;
                const uint32_t po = wrap ? pl0n : pl0;
                const bool pok = wrap ? okn : ok;
                if (RAG) {
                    nB.load_s(Bv + (int64_t)nn * p.B_dstate_stride, (int32_t)po, pok, REV ? L - K : 0);
                    nC.load_s(Cv + (int64_t)nn * p.C_dstate_stride, (int32_t)po, pok, REV ? L - K : 0);
                } else {
                    nB.load(Bv + (int64_t)nn * p.B_dstate_stride, po, pok);
                    nC.load(Cv + (int64_t)nn * p.C_dstate_stride, po, pok);
                }
            }
            const float An = readlane_f(A_mine, n);
            const float hin = readlane_f(hreg, n);
            const f2 An2 = f2{An, An};
            f2 a2[K / 2], bx2[K / 2];
#pragma unroll
            for (int k = 0; k < K / 2; ++k) {
                const f2 t = dl2[k] * An2;
                a2[k] = f2{fast_exp2(t.x), fast_exp2(t.y)};
                bx2[k] = du2[k] * f2{cB.at(2 * k), cB.at(2 * k + 1)};
            }
            float px = 0.f;
#pragma unroll
            for (int i = 0; i < K; ++i) px = fmaf(VMS_ELP(a2, i), px, VMS_ELP(bx2, i));
            float pa = fast_exp2(sdl * An);  // product of the lane's K a_i
            wave_scan_fused_p(pa, px);
            // exclusive prefix of this lane, seeded with the state carried from earlier chunks
            const float ea = dpp_mov<DPP_WAVE_SHR1, 0xf>(1.f, pa);
            const float ex = dpp_mov<DPP_WAVE_SHR1, 0xf>(0.f, px);
            float xs = fmaf(ea, hin, ex);
            const float hend = fmaf(pa, hin, px);  // state after this lane's last element
            if (p.x_has_sub == 1 && ((lane + 1) * K) % 128 == 0) {  // 128-element sub-checkpoints for the backward kernel
                const int i128 = (c * CS + (lane + 1) * K) / 128 - 1;
                xck[(int64_t)(i128 >> 4) * xpitch + 2 * N + (i128 & 15) * N + n] = hend;
            }
            const float hout = readlane_f(hend, 63);
            if (lane == n) hreg = hout;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                xs = fmaf(VMS_ELP(a2, i), xs, VMS_ELP(bx2, i));
                VMS_ELP(bx2, i) = xs;  // x_i
            }
            // 8-element checkpoints (vms_hip.h, x_has_sub == 3): the state after the lane's 8th and 16th element
            if (p.x_has_sub == 3 && ok) {
                float* const xq = xck + (int64_t)(c >> 1) * xpitch + 2 * N + (((n >> 2) * 256 + (c & 1) * 128 + 2 * lane) * 4 + (n & 3));
                xq[0] = bx2[3].y;
                xq[4] = bx2[7].y;
            }
#pragma unroll
            for (int k = 0; k < K / 2; ++k) y2[k] = pk_fma_p(f2{cC.at(2 * k), cC.at(2 * k + 1)}, bx2[k], y2[k]);
        };
#pragma unroll 1
        for (int n = 0; n < N; n += 2) {
            do_state(n, rB0, rC0, rB1, rC1);
            do_state(n + 1, rB1, rC1, rB0, rC0);
        }
        float y[K];
#pragma unroll
        for (int i = 0; i < K; ++i) y[i] = VMS_ELP(y2, i);
        if (full) {
            if (ok) store_p<T, REV>(out_b + (o_out + pl0), y);
        } else {
            store_partial_p<T, REV>(out_b + o_out, l0, L, nv, y);
        }
        if (HZ) {
            RawP<T, REV> tz;
            if (full) tz.load(z_b, o_z + pl0, ok);
            else tz.load_partial(z_b + o_z, l0, L, nv);
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const float zv = tz.at(i);
                y[i] *= zv * sigmoidf_(zv);
            }
            if (p.out_z_accumulate) {  // out_z += (vms_hip.h); loaded only now: the kernel sits at 128 VGPRs
                RawP<T, REV> told;
                if (full) told.load(outz_b, o_oz + pl0, ok);
                else told.load_partial(outz_b + o_oz, l0, L, nv);
#pragma unroll
                for (int i = 0; i < K; ++i) y[i] += told.at(i);
            }
            if (full) {
                if (ok) store_p<T, REV>(outz_b + (o_oz + pl0), y);
            } else {
                store_partial_p<T, REV>(outz_b + o_oz, l0, L, nv, y);
            }
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


// =====================================================================================================================
// Same kernel with B / C shared by the workgroup as fp32 in LDS (whole-vector rows: seqlen % 16 == 0).
//
// The SQ counters of the kernel above (profiles/r02a_sq_scan.md) show a VALU-issue-bound loop: 125 VALU
// instructions per state and wave, of which 32 only widen the bf16 B / C of the state -- the same 2 x 1024 values
// in every wave of a batch.  Here a workgroup = 8 rows of one (batch, group); B / C travel global -> registers ->
// fp32 -> LDS once per workgroup, four states at a time ([tensor][state % 4][1024] = 32 KB, two such buffers), one
// group ahead of the compute; a lane then takes its 16 + 16 values of a state with eight ds_read_b128 (positions
// stored as [i / 4][lane][i % 4], so consecutive lanes hit consecutive 16-byte slots).  Per state and wave: 4
// widening ops + 1 ds_write_b128 instead of 32 widening ops; one workgroup barrier per four states.  64 KB of LDS
// per workgroup: two workgroups = 16 waves per CU, the same 4 waves per SIMD as before.
// timing-only ablations (wrong results): tools/variant.sh <tag> -DVMS_ABL_FWD_NOBAR=1 / -DVMS_ABL_FWD_NOSTAGE=1
#ifndef VMS_ABL_FWD_NOBAR
#define VMS_ABL_FWD_NOBAR 0
#endif
#ifndef VMS_ABL_FWD_NOCKST
#define VMS_ABL_FWD_NOCKST 0   /* no checkpoint stores (the LDS park and its reads stay) */
#endif
#ifndef VMS_ABL_FWD_NOPARK
#define VMS_ABL_FWD_NOPARK 0   /* no LDS park: the stores write register values */
#endif
#ifndef VMS_ABL_FWD_NOSTAGE
#define VMS_ABL_FWD_NOSTAGE 0
#endif
#ifndef VMS_X8_AUX
#define VMS_X8_AUX 2
#endif
constexpr int kX8Aux = VMS_X8_AUX;                    // cache policy of the checkpoint stores: nt (streaming); none of the bits changes their cost
constexpr int kLW = 8;                       // waves (rows) per workgroup
constexpr int kLG = 4;                       // states per staged group
constexpr int kLGroupFloats = 2 * kLG * kWave * kPK;   // [tensor][state % 4][1024]
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) f4 lds_f4p;

template <typename T, bool REV>
struct Raw8 {
    static constexpr int EPV = 16 / sizeof(T);
    vec_t<T, EPV> v[8 / EPV];
    __device__ __forceinline__ void load(const T* __restrict__ base, uint32_t off, bool valid) {
        const vec_t<T, EPV>* vp = reinterpret_cast<const vec_t<T, EPV>*>(base + (valid ? off : 0u));
#pragma unroll
        for (int i = 0; i < 8 / EPV; ++i) v[i] = vp[i];
    }
    __device__ __forceinline__ float at(int i) const {
        const int e = REV ? 7 - i : i;
        return static_cast<float>(v[e / EPV][e % EPV]);
    }
};

// XC: the state after every 8 elements goes to x (vms_hip.h x_has_sub == 3)
// TQ: the instantiation that also holds the four-states-at-a-time form of a row's last chunk (launched only for rows that have
// such a chunk: with the form compiled in, the whole-wave loop of (8, 1024, 8192) ran 2 % slower -- one more spilled register)
// NT: dstate as a compile-time value (16: the tuned instantiation) or 0 = p.dstate in {4, 8} read at run time (round 6: the suite's
// d_state = 4 model, avion/models/model_clip.py:945-947, ran on the generic kernels).  Everything N decides sits outside the
// per-state body: the staged groups per chunk (N / 4), the checkpoint offsets, which lanes hold A / the running state.
template <typename T, bool HZ, bool REV, bool XC, bool TQ, int NT>
__device__ __forceinline__ void scan_fwd_lds_body(const vms_scan_fwd_params& p, const int n_seg, const float2* __restrict__ seg_carry) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int K = kPK, CS = kWave * K;
    const int N = NT ? NT : p.dstate;
    const int NG = N / kLG;                                   // staged groups of 4 states per chunk: 4, 2 or 1
    const int ng_sh = N == 16 ? 2 : N == 8 ? 1 : 0;           // log2(NG)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wg_per_seg = gridDim.x / n_seg;
    const int seg = blockIdx.x / wg_per_seg, wg = blockIdx.x - seg * wg_per_seg;
    const int b = wg % p.batch;
    const int d0 = (wg / p.batch) * kLW;
    const int d = d0 + wave;
    const bool row_ok = d < p.dim;           // rows past the end still stage B / C and meet the barriers
    const int dc = row_ok ? d : p.dim - 1;
    const int g = d0 / (p.dim / p.n_groups); // host guarantees one group per workgroup
    const int L = p.seqlen;

    const T* const u_b = static_cast<const T*>(p.u);
    const T* const dt_b = static_cast<const T*>(p.delta);
    T* const out_b = static_cast<T*>(p.out);
    const T* const z_b = static_cast<const T*>(p.z);
    T* const outz_b = static_cast<T*>(p.out_z);
    const uint32_t o_u = static_cast<uint32_t>((int64_t)b * p.u_batch_stride + (int64_t)dc * p.u_d_stride);
    const uint32_t o_dt = static_cast<uint32_t>((int64_t)b * p.delta_batch_stride + (int64_t)dc * p.delta_d_stride);
    const uint32_t o_out = static_cast<uint32_t>((int64_t)b * p.out_batch_stride + (int64_t)dc * p.out_d_stride);
    const uint32_t o_z = HZ ? static_cast<uint32_t>((int64_t)b * p.z_batch_stride + (int64_t)dc * p.z_d_stride) : 0u;
    const uint32_t o_oz = HZ ? static_cast<uint32_t>((int64_t)b * p.out_z_batch_stride + (int64_t)dc * p.out_z_d_stride) : 0u;
    const T* const Bv = static_cast<const T*>(p.B) + (int64_t)b * p.B_batch_stride + (int64_t)g * p.B_group_stride;
    const T* const Cv = static_cast<const T*>(p.C) + (int64_t)b * p.C_batch_stride + (int64_t)g * p.C_group_stride;
    const int64_t xpitch = p.x_chunk_stride ? p.x_chunk_stride : 2 * N;
    float* const xck = static_cast<float*>(p.x) + ((int64_t)b * p.dim + dc) * p.n_chunks * xpitch;
    const float Dd = p.D ? static_cast<const float*>(p.D)[dc] : 0.f;
    const float bias = p.delta_bias ? static_cast<const float*>(p.delta_bias)[dc] : 0.f;
    const float A_mine = static_cast<const float*>(p.A)[(int64_t)dc * p.A_d_stride + (int64_t)(lane & (N - 1)) * p.A_dstate_stride] * kLog2e;
    float hreg = 0.f;

    const int n_kchunks = (L + CS - 1) / CS;
    const int cps = (n_kchunks + n_seg - 1) / n_seg;
    const int c_lo = seg * cps, c_hi = (c_lo + cps < n_kchunks) ? c_lo + cps : n_kchunks;
    if (seg > 0 && lane < N) {
        const float2* cp = seg_carry + (((int64_t)b * p.dim + dc) * n_seg) * N + lane;
        for (int s2 = 0; s2 < seg; ++s2) {
            const float2 pq = cp[(int64_t)s2 * N];
            hreg = fmaf(pq.x, hreg, pq.y);
        }
    }
    // staging: a group = 4 states x 2 tensors x 1024 positions = 1024 pieces of 8 values; thread t owns pieces t and
    // t + 512: tensor pid >> 9, state (pid >> 7) & 3, positions 8 (pid & 127) .. + 7 of the chunk
    Raw8<T, REV> stg[2];
    bool st_ok[2] = {false, false};
    auto stage_issue = [&](int gi) __attribute__((always_inline)) {   // gi = NG chunk + state group, global order
        const int cc = gi >> ng_sh, n0 = (gi & (NG - 1)) * kLG;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int pid = (int)threadIdx.x + kLW * kWave * h;
            const int ten = pid >> 9, n = n0 + ((pid >> 7) & 3), l8 = cc * CS + (pid & 127) * 8;
            st_ok[h] = cc < c_hi && l8 < L;
            const T* src = ten ? Cv + (int64_t)n * p.C_dstate_stride : Bv + (int64_t)n * p.B_dstate_stride;
            stg[h].load(src, REV ? L - l8 - 8 : l8, st_ok[h]);
        }
    };
    auto stage_commit = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int pid = (int)threadIdx.x + kLW * kWave * h;
            const int p8 = pid & 127;       // positions 8 p8 .. + 7 = lane p8 / 2, elements 8 (p8 & 1) .. + 7
            f4 lo, hi;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                lo[i] = st_ok[h] ? stg[h].at(i) : 0.f;
                hi[i] = st_ok[h] ? stg[h].at(4 + i) : 0.f;
            }
            lds_f4p* dst = (lds_f4p*)(smem + buf * kLGroupFloats + (pid >> 7) * (kWave * K)) + (2 * (p8 & 1)) * kWave + (p8 >> 1);
            dst[0] = lo;
            dst[kWave] = hi;
        }
    };
    typedef uint32_t u32x4_p __attribute__((ext_vector_type(4)));
    // per row (wave-uniform); x rows of >= 2 GiB do not exist (n_chunks * pitch floats)
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(xck, 0, (int)(p.n_chunks * xpitch * 4), 0x00020000);
    typedef __attribute__((address_space(3))) float lds_f1p;
    lds_f1p* const park = (lds_f1p*)(smem + 2 * kLGroupFloats) + wave * (2 * kLG * kWave) + lane;   // [wave][8th | 16th][state % 4][lane]
    // 8-element checkpoints of group g of chunk cc, out of the park: two 16-byte stores per lane through the row's buffer
    // resource (uniform offsets in the SGPR operand; lanes past the end, and everything when !valid, dropped by an out-of-range
    // offset: no branch, so that the compiler can count what is in flight).  Layout [state / 4][8-element index][state % 4];
    // the park doubles as a transposition: lane l stores index l (source lane l / 2, its 8th or 16th element) and index
    // 64 + l, so that each instruction writes 1 KB of whole lines (16 bytes at a 32-byte stride -- a lane storing its own two
    // indices -- cost +90 us per launch).
    // WHEN: memory operations complete in issue order, loads and stores alike (one vmcnt).  Stored where the values appear, or
    // right after the group's barrier, the stores sit in front of the next B / C requests and every wait for those is a wait
    // for the stores' round trip (+35-50 us per launch).  They go out right BEHIND the next group's B / C requests instead:
    // the wait for B / C becomes vmcnt(2) and leaves them in flight.
    auto flush_park = [&](const int cc, const int g, const bool valid) __attribute__((always_inline)) {
        const int so = (int)(((cc >> 1) * xpitch + 2 * N + (g * 256 + (cc & 1) * 128) * 4) * 4);
        const lds_f1p* const src = park - lane + (lane & 1) * (kLG * kWave) + (lane >> 1);
        f4 va, vb;
#pragma unroll
        for (int s4 = 0; s4 < kLG; ++s4) {
            va[s4] = VMS_ABL_FWD_NOPARK ? hreg + (float)s4 : src[s4 * kWave];
            vb[s4] = VMS_ABL_FWD_NOPARK ? hreg - (float)s4 : src[s4 * kWave + 32];
        }
        const int la = cc * CS + (lane >> 1) * K;
        const uint32_t voa = valid && la < L && row_ok ? (uint32_t)lane * 16u : 0x80000000u;
        const uint32_t vob = valid && la + 32 * K < L && row_ok ? (uint32_t)lane * 16u : 0x80000000u;
        if (!VMS_ABL_FWD_NOCKST) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_p, va), xrs, voa, so, kX8Aux);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_p, vb), xrs, vob, so + 1024, kX8Aux);
        }
        // the first dword of the second store's data came out overwritten now and then (3-8 values per 65 K) when hipcc put
        // a VALU write of that register right behind the store (the 16-byte-store write-data hazard, which its hazard
        // recogniser takes to be absent with an SGPR offset): the data registers stay live up to two wait states behind the
        // stores (a scheduling barrier here instead spilled 40 registers into the state loop)
        asm volatile("s_nop 1" ::"v"(va), "v"(vb));
    };
    int gi = c_lo * NG;
    stage_issue(gi);
    stage_commit(0);
    __syncthreads();
    // the chunk body twice, as a whole-wave pass (the loop) and as the tail form (once, behind the loop): two separate code
    // regions, so that the tail form's temporaries do not sit in the loop's register budget (inside the loop, as a branch, they
    // pushed it into scratch: 2.5x slower)
    auto run_chunk = [&](const int c, auto tail_tag) __attribute__((always_inline)) {
        // A row's last chunk of <= 256 elements (L = 3136: 64, L = 2304: 256) would cost a whole 1024-element pass with 4 - 16 of
        // the 64 lanes at work.  It runs FOUR STATES AT A TIME instead: DPP row g (16 lanes x 16 elements = the 256 positions)
        // carries state 4 sg + g, the lane aggregates are scanned per row, the four rows' y are summed at the end (do_quad):
        // a quarter of a pass.  (8, 768, 3136): 4 -> 3.25 passes per row.
        constexpr bool tail = decltype(tail_tag)::value;
        const int li = tail ? (lane & 15) : lane;           // the lane's slot of 16 elements inside the chunk
        const int l0 = c * CS + li * K;
        const bool ok = l0 < L && row_ok;
        const uint32_t pl0 = REV ? L - l0 - K : l0;
        f2 dl2[K / 2], du2[K / 2], y2[K / 2];
        float sdl = 0.f;
        {
            RawP<T, REV> t0, t1;
            t0.load_row(u_b, o_u + pl0, ok);
            t1.load_row(dt_b, o_dt + pl0, ok);
            // per element-PAIR (round 3): packed adds / multiplies around the transcendentals
            const float Dq = tail && lane >= 16 ? 0.f : Dd;   // tail: the four rows' y are summed, D u counts once
            const f2 bias2 = f2{bias, bias}, Dd2 = f2{Dq, Dq};
            f2 sd2 = f2{0.f, 0.f};
#pragma unroll
            for (int k = 0; k < K / 2; ++k) {
                f2 t = f2{t1.at(2 * k), t1.at(2 * k + 1)} + bias2;
                if (p.delta_softplus) t = softplus2_p(t);
                t = ok ? t : f2{0.f, 0.f};  // past the end: delta = 0 -> a = 1, b = 0 (identity)
                const f2 uv = f2{t0.at(2 * k), t0.at(2 * k + 1)};
                dl2[k] = t;
                du2[k] = t * uv;
                y2[k] = Dd2 * uv;
                sd2 = sd2 + t;
            }
            sdl = sd2.x + sd2.y;
        }
        auto do_state = [&](const int n, const int buf) __attribute__((always_inline)) {
            const lds_f4p* bsrc = (const lds_f4p*)(smem + buf * kLGroupFloats + (n & 3) * (kWave * K)) + lane;
            const lds_f4p* csrc = bsrc + kLG * (kWave * K) / 4;
            const f4 b0 = bsrc[0], b1 = bsrc[kWave], b2 = bsrc[2 * kWave], b3 = bsrc[3 * kWave];
            const float An = readlane_f(A_mine, n);
            const float hin = readlane_f(hreg, n);
            const f2 An2 = f2{An, An};
            f2 a2[K / 2], bx2[K / 2];
#pragma unroll
            for (int k = 0; k < K / 2; ++k) {
                const f4 bq = k < 2 ? b0 : k < 4 ? b1 : k < 6 ? b2 : b3;
                const f2 t = dl2[k] * An2;
                a2[k] = f2{fast_exp2(t.x), fast_exp2(t.y)};
                bx2[k] = du2[k] * ((k & 1) ? f2{bq.z, bq.w} : f2{bq.x, bq.y});
            }
            float px = 0.f;
#pragma unroll
            for (int i = 0; i < K; ++i) px = fmaf(VMS_ELP(a2, i), px, VMS_ELP(bx2, i));
            float pa = fast_exp2(sdl * An);  // product of the lane's K a_i
            wave_scan_fused_p(pa, px);
            const f4 c0 = csrc[0], c1 = csrc[kWave], c2 = csrc[2 * kWave], c3 = csrc[3 * kWave];   // arrive during the second chain
            const float ea = dpp_mov<DPP_WAVE_SHR1, 0xf>(1.f, pa);
            const float ex = dpp_mov<DPP_WAVE_SHR1, 0xf>(0.f, px);
            float xs = fmaf(ea, hin, ex);
            const float hend = fmaf(pa, hin, px);  // state after this lane's last element
            if (!XC && p.x_has_sub == 1 && ((lane + 1) * K) % 128 == 0 && row_ok) {  // 128-element sub-checkpoints for the backward kernel
                const int i128 = (c * CS + (lane + 1) * K) / 128 - 1;
                xck[(int64_t)(i128 >> 4) * xpitch + 2 * N + (i128 & 15) * N + n] = hend;
            }
            const float hout = readlane_f(hend, 63);
            if (lane == n) hreg = hout;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                xs = fmaf(VMS_ELP(a2, i), xs, VMS_ELP(bx2, i));
                VMS_ELP(bx2, i) = xs;  // x_i
            }
            // 8-element checkpoints (vms_hip.h, x_has_sub == 3): the state after the lane's 8th and 16th element, parked in
            // a wave-private LDS slot until the next group's B / C requests are out (flush_park)
            if constexpr (XC) {
                if (!VMS_ABL_FWD_NOPARK) {
                    park[(n & 3) * kWave] = bx2[3].y;
                    park[(kLG + (n & 3)) * kWave] = bx2[7].y;
                }
            }
#pragma unroll
            for (int k = 0; k < K / 2; ++k) {
                const f4 cq = k < 2 ? c0 : k < 4 ? c1 : k < 6 ? c2 : c3;
                y2[k] = pk_fma_p((k & 1) ? f2{cq.z, cq.w} : f2{cq.x, cq.y}, bx2[k], y2[k]);
            }
        };
        // the four states of a staged group at once, one per DPP row (tail chunks, see above): do_state with per-row A / state
        auto do_quad = [&](const int sg, const int buf) __attribute__((always_inline)) {
            const int g4 = lane >> 4;
            const lds_f4p* bsrc = (const lds_f4p*)(smem + buf * kLGroupFloats + g4 * (kWave * K)) + li;
            const lds_f4p* csrc = bsrc + kLG * (kWave * K) / 4;
            const f4 b0 = bsrc[0], b1 = bsrc[kWave], b2 = bsrc[2 * kWave], b3 = bsrc[3 * kWave];
            const int n0 = 4 * sg;
            const float A0 = readlane_f(A_mine, n0), A1 = readlane_f(A_mine, n0 + 1), A2 = readlane_f(A_mine, n0 + 2), A3 = readlane_f(A_mine, n0 + 3);
            const float H0 = readlane_f(hreg, n0), H1 = readlane_f(hreg, n0 + 1), H2 = readlane_f(hreg, n0 + 2), H3 = readlane_f(hreg, n0 + 3);
            const float An = g4 == 0 ? A0 : g4 == 1 ? A1 : g4 == 2 ? A2 : A3;
            const float hin = g4 == 0 ? H0 : g4 == 1 ? H1 : g4 == 2 ? H2 : H3;
            const f2 An2 = f2{An, An};
            f2 a2[K / 2], bx2[K / 2];
#pragma unroll
            for (int k = 0; k < K / 2; ++k) {
                const f4 bq = k < 2 ? b0 : k < 4 ? b1 : k < 6 ? b2 : b3;
                const f2 t = dl2[k] * An2;
                a2[k] = f2{fast_exp2(t.x), fast_exp2(t.y)};
                bx2[k] = du2[k] * ((k & 1) ? f2{bq.z, bq.w} : f2{bq.x, bq.y});
            }
            float px = 0.f;
#pragma unroll
            for (int i = 0; i < K; ++i) px = fmaf(VMS_ELP(a2, i), px, VMS_ELP(bx2, i));
            float pa = fast_exp2(sdl * An);
            row_scan_fused_p(pa, px);
            const f4 c0 = csrc[0], c1 = csrc[kWave], c2 = csrc[2 * kWave], c3 = csrc[3 * kWave];
            const float ea = dpp_mov<DPP_ROW_SHR1, 0xf>(1.f, pa);   // (the row's first lane keeps the identity)
            const float ex = dpp_mov<DPP_ROW_SHR1, 0xf>(0.f, px);
            float xs = fmaf(ea, hin, ex);
            const float hend = fmaf(pa, hin, px);
            if (!XC && p.x_has_sub == 1 && ((li + 1) * K) % 128 == 0 && row_ok) {
                const int i128 = (c * CS + (li + 1) * K) / 128 - 1;
                xck[(int64_t)(i128 >> 4) * xpitch + 2 * N + (i128 & 15) * N + n0 + g4] = hend;
            }
            // lanes past the row's end are identity steps: the last lane of DPP row g holds state n0 + g after the chunk
            const float E0 = readlane_f(hend, 15), E1 = readlane_f(hend, 31), E2 = readlane_f(hend, 47), E3 = readlane_f(hend, 63);
            hreg = lane == n0 ? E0 : lane == n0 + 1 ? E1 : lane == n0 + 2 ? E2 : lane == n0 + 3 ? E3 : hreg;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                xs = fmaf(VMS_ELP(a2, i), xs, VMS_ELP(bx2, i));
                VMS_ELP(bx2, i) = xs;
            }
            if constexpr (XC) {   // the slots do_state's lane li would use for state n0 + g
                (park - lane + li)[g4 * kWave] = bx2[3].y;
                (park - lane + li)[(kLG + g4) * kWave] = bx2[7].y;
            }
#pragma unroll
            for (int k = 0; k < K / 2; ++k) {
                const f4 cq = k < 2 ? c0 : k < 4 ? c1 : k < 6 ? c2 : c3;
                y2[k] = pk_fma_p((k & 1) ? f2{cq.z, cq.w} : f2{cq.x, cq.y}, bx2[k], y2[k]);
            }
        };
#pragma unroll 1
        for (int sg = 0; sg < NG; ++sg, ++gi) {
            const int buf = gi & 1;
            if (!VMS_ABL_FWD_NOSTAGE) stage_issue(gi + 1);     // the next group (of the next chunk after the last one) travels while this one computes
            if constexpr (XC) flush_park(sg > 0 ? c : c - 1, (sg + NG - 1) & (NG - 1), sg > 0 || c > c_lo);   // the group before this one
            if constexpr (tail) {
                do_quad(sg, buf);
            } else {
                do_state(4 * sg, buf);
                do_state(4 * sg + 1, buf);
                do_state(4 * sg + 2, buf);
                do_state(4 * sg + 3, buf);
            }
            stage_commit(buf ^ 1);   // every wave left that buffer at the previous barrier
            if (!VMS_ABL_FWD_NOBAR) __syncthreads();
        }
        float y[K];
#pragma unroll
        for (int i = 0; i < K; ++i) y[i] = VMS_ELP(y2, i);
        if constexpr (tail) {   // sum of the four DPP rows' partial y (lane li of each row holds the same 16 positions)
#pragma unroll
            for (int i = 0; i < K; ++i) {
                y[i] += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 16) << 2, __builtin_bit_cast(int, y[i])));
                y[i] += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __builtin_bit_cast(int, y[i])));
            }
#pragma unroll
            for (int k = 0; k < K / 2; ++k) y2[k] = f2{y[2 * k], y[2 * k + 1]};
        }
        const bool st_okk = ok && (!tail || lane < 16);     // tail: DPP row 0 writes the positions
        if (st_okk) store_p<T, REV>(out_b + (o_out + pl0), y);
        if (HZ) {
            RawP<T, REV> tz;
            tz.load_row(z_b, o_z + pl0, ok);
#pragma unroll
            for (int k = 0; k < K / 2; ++k) {
                const f2 g = y2[k] * silu2_p(f2{tz.at(2 * k), tz.at(2 * k + 1)});
                y[2 * k] = g.x;
                y[2 * k + 1] = g.y;
            }
            if (p.out_z_accumulate) {
                RawP<T, REV> told;
                told.load(outz_b, o_oz + pl0, ok);
#pragma unroll
                for (int i = 0; i < K; ++i) y[i] += told.at(i);
            }
            if (st_okk) store_p<T, REV>(outz_b + (o_oz + pl0), y);
        }
        const bool last = c == n_kchunks - 1;
        const int pos = (c + 1) * CS;
        if (lane < N && row_ok && (last || pos % 1024 == 0)) {
            const int blk = last ? (L - 1) / 2048 : (pos - 1) / 2048;
            const int r = (last ? L : pos) - blk * 2048;
            float* xb = xck + (int64_t)blk * xpitch;
            if (r <= 1024) xb[2 * lane] = hreg;
            if (r == 2048 || last) xb[2 * lane + 1] = hreg;
        }
    };
    // (workgroup-uniform) the range's last chunk is the row's last and holds <= 256 elements
    if constexpr (TQ) {
        const bool has_tail = c_hi == n_kchunks && c_hi > c_lo && L - (c_hi - 1) * CS <= 16 * K;
        const int c_main = has_tail ? c_hi - 1 : c_hi;
        for (int c = c_lo; c < c_main; ++c) run_chunk(c, std::false_type{});
        if (has_tail) run_chunk(c_hi - 1, std::true_type{});
    } else {
        for (int c = c_lo; c < c_hi; ++c) run_chunk(c, std::false_type{});
    }
    if constexpr (XC) flush_park(c_hi - 1, NG - 1, c_hi > c_lo);   // the last group of the last chunk
}


// RM: 0 = left-to-right, 1 = right-to-left, 2 = per batch entry (vms_hip.h reverse_from: entries >= p.reverse_from run
// right-to-left).  A workgroup serves one batch entry, so the direction is workgroup-uniform: one branch, both bodies.
template <typename T, bool HZ, int RM, bool XC, bool TQ = false, int NT = kPN>
__global__ __launch_bounds__(kLW* kWave, 4) void scan_fwd_lds_kernel(const vms_scan_fwd_params p, const int n_seg,
                                                                        const float2* __restrict__ seg_carry) {
    if constexpr (RM == 2) {
        const int wg_per_seg = gridDim.x / n_seg;
        const int b = (int)(blockIdx.x % wg_per_seg) % p.batch;
        if (b >= p.reverse_from) scan_fwd_lds_body<T, HZ, true, XC, TQ, NT>(p, n_seg, seg_carry);
        else scan_fwd_lds_body<T, HZ, false, XC, TQ, NT>(p, n_seg, seg_carry);
    } else {
        scan_fwd_lds_body<T, HZ, RM == 1, XC, TQ, NT>(p, n_seg, seg_carry);
    }
}


// =====================================================================================================================
// Few rows, long sequences in ONE pass (round 4): the 16 states of a row spread over the 4 waves of a workgroup.
//
// With fewer rows than SIMDs -- batch 1, 768 channels, 65,536 tokens, BASELINE configs[4] -- the kernels above cut every row
// into ranges of chunks and pay a carry pass for it: one more exp per (element, state), ~60 % of the main pass, whatever the
// range count (profiles/r04_seg_sweep.txt: 446-535 us against ~290 for the main pass alone).  States are the other axis of
// parallelism and need no second pass: a workgroup = ONE row, wave w = states 4w .. 4w+3.  What the states share is done
// once per workgroup and passed through LDS:
//   * prologue: wave w turns elements [256 w, 256 w + 256) of the 1024-element chunk (4 per lane, contiguous 512 B per wave)
//     into delta = softplus(delta_raw + bias) and delta * u -> LDS; after a barrier every wave reads its lane's 16 + 16 values;
//   * its four states exactly as scan_fwd_pair_kernel runs them (per-wave B / C, the next state's -- after the last: the next
//     chunk's first -- requested while the current one computes);
//   * y: every wave leaves its partial sum over 4 states in LDS; after a second barrier wave w adds the four partials of ITS
//     256 elements, D u, the z gate, and stores them.
// 8-element checkpoints (x_has_sub == 3): a wave's four states are one [state / 4] block of the layout; transposed through a
// wave-private 2 KB park into two 1 KB-contiguous 16-byte stores, like scan_fwd_lds_kernel's.
// LDS 32 KB per workgroup, <= 168 VGPRs: 3 workgroups per CU = 3 waves per SIMD at 768 rows.
constexpr int kSW = 4;                                                            // waves per workgroup = state groups
constexpr int kSgFloats = 2 * 1024 + kSW * 1024 + kSW * 2 * kLG * kWave;          // delta, delta u | y partials | parks
#ifndef VMS_SG_MIN_ROWS
#define VMS_SG_MIN_ROWS 2   /* rows * VMS_SG_MIN_ROWS >= SIMDs: below that even 4 waves per row leave SIMDs empty: ranges */
#endif
#ifndef VMS_SG_MAX_HALVES
#define VMS_SG_MAX_HALVES 3 /* rows <= VMS_SG_MAX_HALVES * SIMDs / 2: (2, 768, 32768) = 1.5 rows per SIMD 456 -> 403 us; at 3 per SIMD a tie */
#endif

template <typename T, bool REV>
struct Raw4 {   // 4 consecutive logical elements (REV: stored right-to-left)
    vec_t<T, 4> v;
    __device__ __forceinline__ void load(const T* __restrict__ base, uint32_t off, bool valid) {
        v = *reinterpret_cast<const vec_t<T, 4>*>(base + (valid ? off : 0u));
    }
    __device__ __forceinline__ float at(int i) const { return static_cast<float>(v[REV ? 3 - i : i]); }
};
template <typename T, bool REV>
__device__ __forceinline__ void store4_p(T* __restrict__ ptr, const float (&in)[4]) {
    vec_t<T, 4> t;
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = static_cast<T>(in[REV ? 3 - e : e]);
    *reinterpret_cast<vec_t<T, 4>*>(ptr) = t;
}
__device__ __forceinline__ void lds_barrier_p() {   // orders LDS traffic only: loads / stores in flight stay in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <typename T, bool HZ, bool REV>
__device__ __forceinline__ void scan_fwd_sg_body(const vms_scan_fwd_params& p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int K = kPK, N = kPN, CS = kWave * K;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x % p.batch;
    const int d = blockIdx.x / p.batch;                  // grid = batch * dim: one row per workgroup
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
    // lane n (< 16) keeps A[d][n] * log2(e) and the running state of recurrence n; this wave advances lanes 4w .. 4w+3
    const float A_mine = static_cast<const float*>(p.A)[(int64_t)d * p.A_d_stride + (int64_t)(lane & 15) * p.A_dstate_stride] * kLog2e;
    float hreg = 0.f;

    lds_f4p* const D4 = (lds_f4p*)smem;                                // [delta | delta u][1024] by logical position in the chunk
    lds_f4p* const Y4 = (lds_f4p*)(smem + 2 * 1024);                   // [wave][1024]
    lds_f4p* const park4 = (lds_f4p*)(smem + 2 * 1024 + kSW * 1024) + w * (2 * kWave);   // [8-element index (128)][state % 4]
    typedef uint32_t u32x4_p __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(xck, 0, (int)(p.n_chunks * xpitch * 4), 0x00020000);

    const int n_k = (L + CS - 1) / CS;
    const int q_off = 256 * w + 4 * lane;                 // this lane's 4 prologue / epilogue elements inside a chunk
    Raw4<T, REV> ru, rdt, rz, rold;                       // ... of the NEXT chunk, in flight while the current one computes
    auto request_q = [&](int c) __attribute__((always_inline)) {
        const int lq = c * CS + q_off;
        const bool v = c < n_k && lq < L;
        const uint32_t pq = REV ? L - lq - 4 : lq;
        ru.load(u_b, o_u + pq, v);
        rdt.load(dt_b, o_dt + pq, v);
        if (HZ) {
            rz.load(z_b, o_z + pq, v);
            if (p.out_z_accumulate) rold.load(outz_b, o_oz + pq, v);
        }
    };
    RawP<T, REV> rB0, rC0, rB1, rC1;
    const int n_first = 4 * w;
    request_q(0);
    {
        const int l0 = lane * K;
        rB0.load(Bv + (int64_t)n_first * p.B_dstate_stride, REV ? L - l0 - K : l0, l0 < L);
        rC0.load(Cv + (int64_t)n_first * p.C_dstate_stride, REV ? L - l0 - K : l0, l0 < L);
    }
    for (int c = 0; c < n_k; ++c) {
        const int l0 = c * CS + lane * K;
        const bool ok = l0 < L;
        const uint32_t pl0 = REV ? L - l0 - K : l0;
        const bool okn = l0 + CS < L;
        const uint32_t pl0n = REV ? L - l0 - CS - K : l0 + CS;
        const int lq = c * CS + q_off;
        const bool qok = lq < L;
        const uint32_t pq = REV ? L - lq - 4 : lq;
        // ---- prologue of this wave's quarter -> LDS
        float Du4[4];
        Raw4<T, REV> zc = rz, oldc = rold;
        {
            f4 dl4, du4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float t = rdt.at(i) + bias;
                if (p.delta_softplus) t = softplusf_(t);
                t = qok ? t : 0.f;                      // past the end: delta = 0 -> a = 1, b = 0 (identity)
                const float uv = ru.at(i);
                dl4[i] = t;
                du4[i] = t * uv;
                Du4[i] = Dd * uv;
            }
            D4[q_off / 4] = dl4;
            D4[256 + q_off / 4] = du4;
        }
        request_q(c + 1);
        lds_barrier_p();
        f2 dl2[K / 2], du2[K / 2], y2[K / 2];
        float sdl = 0.f;
        {
            f2 sd2 = f2{0.f, 0.f};
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const f4 a = D4[4 * lane + v], bq = D4[256 + 4 * lane + v];
                dl2[2 * v] = f2{a.x, a.y};
                dl2[2 * v + 1] = f2{a.z, a.w};
                du2[2 * v] = f2{bq.x, bq.y};
                du2[2 * v + 1] = f2{bq.z, bq.w};
                sd2 = sd2 + dl2[2 * v] + dl2[2 * v + 1];
            }
            sdl = sd2.x + sd2.y;
#pragma unroll
            for (int k = 0; k < K / 2; ++k) y2[k] = f2{0.f, 0.f};
        }
        float ck8[kLG], ck16[kLG];
        auto do_state = [&](const int k4, const RawP<T, REV>& cB, const RawP<T, REV>& cC, RawP<T, REV>& nB,
                            RawP<T, REV>& nC) __attribute__((always_inline)) {
            const int n = n_first + k4;
            {   // B / C of the next state -- after this wave's last state: its first state of the next chunk
                const bool wrap = k4 == kLG - 1;
                const int nn = wrap ? n_first : n + 1;
                const uint32_t po = wrap ? pl0n : pl0;
                const bool pok = wrap ? okn : ok;
                nB.load(Bv + (int64_t)nn * p.B_dstate_stride, po, pok);
                nC.load(Cv + (int64_t)nn * p.C_dstate_stride, po, pok);
            }
            const float An = readlane_f(A_mine, n);
            const float hin = readlane_f(hreg, n);
            const f2 An2 = f2{An, An};
            f2 a2[K / 2], bx2[K / 2];
#pragma unroll
            for (int k = 0; k < K / 2; ++k) {
                const f2 t = dl2[k] * An2;
                a2[k] = f2{fast_exp2(t.x), fast_exp2(t.y)};
                bx2[k] = du2[k] * f2{cB.at(2 * k), cB.at(2 * k + 1)};
            }
            float px = 0.f;
#pragma unroll
            for (int i = 0; i < K; ++i) px = fmaf(VMS_ELP(a2, i), px, VMS_ELP(bx2, i));
            float pa = fast_exp2(sdl * An);
            wave_scan_fused_p(pa, px);
            const float ea = dpp_mov<DPP_WAVE_SHR1, 0xf>(1.f, pa);
            const float ex = dpp_mov<DPP_WAVE_SHR1, 0xf>(0.f, px);
            float xs = fmaf(ea, hin, ex);
            const float hend = fmaf(pa, hin, px);
            if (p.x_has_sub == 1 && ((lane + 1) * K) % 128 == 0 && ok) {
                const int i128 = (c * CS + (lane + 1) * K) / 128 - 1;
                xck[(int64_t)(i128 >> 4) * xpitch + 2 * N + (i128 & 15) * N + n] = hend;
            }
            const float hout = readlane_f(hend, 63);
            if (lane == n) hreg = hout;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                xs = fmaf(VMS_ELP(a2, i), xs, VMS_ELP(bx2, i));
                VMS_ELP(bx2, i) = xs;
            }
            ck8[k4] = bx2[3].y;
            ck16[k4] = bx2[7].y;
#pragma unroll
            for (int k = 0; k < K / 2; ++k) y2[k] = pk_fma_p(f2{cC.at(2 * k), cC.at(2 * k + 1)}, bx2[k], y2[k]);
        };
        do_state(0, rB0, rC0, rB1, rC1);
        do_state(1, rB1, rC1, rB0, rC0);
        do_state(2, rB0, rC0, rB1, rC1);
        do_state(3, rB1, rC1, rB0, rC0);
        // ---- this wave's partial y -> LDS
#pragma unroll
        for (int v = 0; v < 4; ++v)
            Y4[w * 256 + 4 * lane + v] = f4{y2[2 * v].x, y2[2 * v].y, y2[2 * v + 1].x, y2[2 * v + 1].y};
        // ---- 8-element checkpoints of the wave's state group: lane l holds indices 2l (8th) and 2l + 1 (16th element); through the
        // park lane l stores index l and index 64 + l: each store instruction covers 1 KB of whole lines (issued behind the next
        // chunk's B / C requests, so the wait for those leaves the stores in flight)
        if (p.x_has_sub == 3) {
            park4[2 * lane] = f4{ck8[0], ck8[1], ck8[2], ck8[3]};
            park4[2 * lane + 1] = f4{ck16[0], ck16[1], ck16[2], ck16[3]};
            const f4 va = park4[lane], vb = park4[kWave + lane];
            const int so = (int)(((c >> 1) * xpitch + 2 * N + (w * 256 + (c & 1) * 128) * 4) * 4);
            const int la = c * CS + (lane >> 1) * K;
            const uint32_t voa = la < L ? (uint32_t)lane * 16u : 0x80000000u;
            const uint32_t vob = la + 32 * K < L ? (uint32_t)lane * 16u : 0x80000000u;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_p, va), xrs, voa, so, kX8Aux);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_p, vb), xrs, vob, so + 1024, kX8Aux);
            asm volatile("s_nop 1" ::"v"(va), "v"(vb));   // the 16-byte-store write-data hazard (see scan_fwd_lds_body)
        }
        lds_barrier_p();
        // ---- epilogue of this wave's quarter: y = sum of the four partials + D u, the z gate, the stores
        {
            const f4 s0 = Y4[q_off / 4], s1 = Y4[256 + q_off / 4], s2 = Y4[512 + q_off / 4], s3 = Y4[768 + q_off / 4];
            float y[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i] = ((s0[i] + s1[i]) + (s2[i] + s3[i])) + Du4[i];
            if (qok) store4_p<T, REV>(out_b + (o_out + pq), y);
            if (HZ) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float zv = zc.at(i);
                    y[i] *= zv * sigmoidf_(zv);
                }
                if (p.out_z_accumulate) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) y[i] += oldc.at(i);
                }
                if (qok) store4_p<T, REV>(outz_b + (o_oz + pq), y);
            }
        }
        // reference-shaped checkpoints every 1024 elements (vms_hip.h); lanes 4w .. 4w+3 of this wave hold its states
        const bool last = c == n_k - 1;
        const int pos = (c + 1) * CS;
        if (lane < N && (lane >> 2) == w && (last || pos % 1024 == 0)) {
            const int blk = last ? (L - 1) / 2048 : (pos - 1) / 2048;
            const int r = (last ? L : pos) - blk * 2048;
            float* xb = xck + (int64_t)blk * xpitch;
            if (r <= 1024) xb[2 * lane] = hreg;
            if (r == 2048 || last) xb[2 * lane + 1] = hreg;
        }
    }
}

template <typename T, bool HZ, int RM>   // RM as for scan_fwd_lds_kernel
__global__ __launch_bounds__(kSW* kWave, 3) void scan_fwd_sg_kernel(const vms_scan_fwd_params p) {
    if constexpr (RM == 2) {
        const int b = (int)blockIdx.x % p.batch;
        if (b >= p.reverse_from) scan_fwd_sg_body<T, HZ, true>(p);
        else scan_fwd_sg_body<T, HZ, false>(p);
    } else {
        scan_fwd_sg_body<T, HZ, RM == 1>(p);
    }
}

// rows between SIMDs / 2 and 1.5 SIMDs (4 SIMDs per CU): too few waves for the row-per-wave kernels, enough for >= 2 waves per
// SIMD here (profiles/r04_sg_probe.txt: 256 rows a tie with 16 ranges, 128 rows 94 vs 160 us for the ranges); rows of >= 4
// chunks (short rows -- the DBM block's 2,304 -- lose: 52 -> 72 us); whole-vector rows; one buffer resource per row
bool scan_fwd_sg_wanted(const vms_scan_fwd_params& p) {
    if (p.seqlen % kPK != 0 || p.x_has_sub == 2 || p.segments >= 1 || p.dtype == VMS_F32) return false;   // (fp32 rows: 64 more VGPRs of B / C)
    if (p.dstate != kPN) return false;                                                                      // one wave per 4 states of 16
    const int64_t rows = (int64_t)p.batch * p.dim, simds = 4 * (int64_t)device_cu_count();
    if (2 * rows > VMS_SG_MAX_HALVES * simds || rows * VMS_SG_MIN_ROWS < simds) return false;
#ifndef VMS_SG_ANY_LEN
    if (p.seqlen < 4 * kWave * kPK) return false;          // short rows: nothing to split anyway
#endif
    return (int64_t)p.n_chunks * (p.x_chunk_stride ? p.x_chunk_stride : 2 * kPN) * 4 < ((int64_t)1 << 31);
}

// ---- state carries of a sequence-split forward --------------------------------------------------------------------
// (P, q) per (row, state) and range of chunks: the state leaving the range is P x_in + q, P = exp2(A sum(delta)),
// q = the recurrence run from x = 0.  The forward kernel without its C / y / z half (~60 % of its work).
template <typename T, bool REV>
__global__ __launch_bounds__(kPRows* kWave, VMS_PAIR_MINWAVES) void scan_fwd_carry_kernel(const vms_scan_fwd_params p,
                                                                                           const int n_seg,
                                                                                           float2* __restrict__ seg_carry) {
    constexpr int K = kPK, N = kPN, CS = kWave * K;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wg_per_seg = gridDim.x / (n_seg - 1);
    const int seg = blockIdx.x / wg_per_seg, wg = blockIdx.x - seg * wg_per_seg;   // ranges 0 .. n_seg-2
    const int b = wg % p.batch;
    const int d = (wg / p.batch) * kPRows + wave;
    if (d >= p.dim) return;
    const int g = d / (p.dim / p.n_groups);
    const int L = p.seqlen;
    const T* const u_b = static_cast<const T*>(p.u);
    const T* const dt_b = static_cast<const T*>(p.delta);
    const uint32_t o_u = static_cast<uint32_t>((int64_t)b * p.u_batch_stride + (int64_t)d * p.u_d_stride);
    const uint32_t o_dt = static_cast<uint32_t>((int64_t)b * p.delta_batch_stride + (int64_t)d * p.delta_d_stride);
    const T* const Bv = static_cast<const T*>(p.B) + (int64_t)b * p.B_batch_stride + (int64_t)g * p.B_group_stride;
    const float bias = p.delta_bias ? static_cast<const float*>(p.delta_bias)[d] : 0.f;
    const float A_mine = static_cast<const float*>(p.A)[(int64_t)d * p.A_d_stride + (int64_t)(lane & 15) * p.A_dstate_stride] * kLog2e;
    float hreg = 0.f, sdl_acc = 0.f;
    const int n_kchunks = (L + CS - 1) / CS;
    const int cps = (n_kchunks + n_seg - 1) / n_seg;
    const int c_lo = seg * cps, c_hi = (c_lo + cps < n_kchunks) ? c_lo + cps : n_kchunks;
    RawP<T, REV> rB0, rB1;
    for (int c = c_lo; c < c_hi; ++c) {
        const int l0 = c * CS + lane * K;
        const bool ok = l0 < L;                       // seqlen % K == 0 (host)
        const uint32_t pl0 = REV ? L - l0 - K : l0;
        const bool okn = l0 + CS < L;
        const uint32_t pl0n = REV ? L - l0 - CS - K : l0 + CS;
        if (c == c_lo) rB0.load(Bv, pl0, ok);
        f2 dl2[K / 2], du2[K / 2];
        float sdl = 0.f;
        {
            RawP<T, REV> t0, t1;
            t0.load(u_b, o_u + pl0, ok);
            t1.load(dt_b, o_dt + pl0, ok);
#pragma unroll
            for (int i = 0; i < K; ++i) {
                float t = t1.at(i) + bias;
                if (p.delta_softplus) t = softplusf_(t);
                t = ok ? t : 0.f;
                dl2[i / 2][i % 2] = t;
                du2[i / 2][i % 2] = t * t0.at(i);
                sdl += t;
            }
        }
        sdl_acc += sdl;
        auto do_state = [&](const int n, const RawP<T, REV>& cB, RawP<T, REV>& nB) __attribute__((always_inline)) {
            {
                const int nn = (n + 1) & (N - 1);
                const bool wrap = n + 1 == N;
                nB.load(Bv + (int64_t)nn * p.B_dstate_stride, wrap ? pl0n : pl0, wrap ? okn : ok);
            }
            const float An = readlane_f(A_mine, n);
            const float hin = readlane_f(hreg, n);
            const f2 An2 = f2{An, An};
            float px = 0.f;
#pragma unroll
            for (int k = 0; k < K / 2; ++k) {
                const f2 t = dl2[k] * An2;
                const f2 a = f2{fast_exp2(t.x), fast_exp2(t.y)};
                const f2 bx = du2[k] * f2{cB.at(2 * k), cB.at(2 * k + 1)};
                px = fmaf(a.x, px, bx.x);
                px = fmaf(a.y, px, bx.y);
            }
            float pa = fast_exp2(sdl * An);
            wave_scan_fused_p(pa, px);
            const float hout = readlane_f(fmaf(pa, hin, px), 63);
            if (lane == n) hreg = hout;
        };
#pragma unroll 1
        for (int n = 0; n < N; n += 2) {
            do_state(n, rB0, rB1);
            do_state(n + 1, rB1, rB0);
        }
    }
    const float tot = wave_sum(sdl_acc);
    if (lane < N) seg_carry[(((int64_t)b * p.dim + d) * n_seg + seg) * N + lane] = float2{fast_exp2(A_mine * tot), hreg};
}

// ranges a forward is split into (1 = not split): only when the rows do not give every SIMD a wave (4 SIMDs per CU);
// aims at ~12 waves per SIMD = three rounds of the 4 resident ones (round 4, profiles/r04_seg_sweep.txt: (1, 768, 65536) 535 us
// with the 4 ranges the earlier 2.5-waves rule gave, 492 / 453 / 464 / 446 us with 6 / 8 / 12 / 16: the carry pass costs the
// same whatever the count, the main pass needs the occupancy the kernel was built for), at least 4 chunks (4096 elements)
// per range; p.segments >= 1 forces a count (vms_hip.h)
int scan_fwd_pair_segments(const vms_scan_fwd_params& p) {
    if (p.seqlen % kPK != 0 || p.dstate != kPN) return 1;   // (dstate 4 / 8: the LDS kernel unsplit; the carry kernel is built for 16)
    if (scan_fwd_sg_wanted(p)) return 1;   // served in one pass by scan_fwd_sg_kernel: no ranges, no workspace
    const int n_k = (p.seqlen + kWave * kPK - 1) / (kWave * kPK);
    const int64_t waves = (int64_t)p.batch * p.dim;
    const int64_t simds = 4 * (int64_t)device_cu_count();
    int want = 1;
    if (p.segments >= 1) {
        want = p.segments;
    } else if (waves <= simds) {
        want = (int)((simds * 12 + waves - 1) / waves);
        if (want > n_k / 4) want = n_k / 4;
    }
    if (want > 16) want = 16;
    if (want > n_k) want = n_k;
    if (want < 2) return 1;
    const int cps = (n_k + want - 1) / want;
    return (n_k + cps - 1) / cps;
}

int64_t scan_fwd_pair_ws_bytes(const vms_scan_fwd_params& p) {
    return (int64_t)p.batch * p.dim * 16 * kPN * (int64_t)sizeof(float2);
}

// 16-byte vector accesses need no alignment on gfx950 (measured: global and buffer dwordx4 at 2-byte aligned
// addresses), so `vec` (16-byte aligned bases and strides) is not required; ragged lengths need readable B / C padding
bool scan_fwd_pair_eligible(const vms_scan_fwd_params& p, bool vec) {
    (void)vec;
    if (!p.is_variable_B || !p.is_variable_C || (p.dstate != kPN && p.dstate != 8 && p.dstate != 4)) return false;
    // dstate 4 / 8 (round 6): the LDS kernel only -- whole-vector rows whose 8-row workgroups share one B / C group
    if (p.dstate != kPN && (p.seqlen % kPK != 0 || p.n_groups < 1 || p.dim % p.n_groups != 0 || (p.dim / p.n_groups) % kLW != 0)) return false;
    if (p.seqlen % kPK != 0 && p.bc_pad < kPK - p.seqlen % kPK) return false;
    const int64_t lim = (int64_t)1 << 31;
    auto span = [&](int64_t bs, int64_t ds) { return (p.batch - 1) * bs + (p.dim - 1) * ds + p.seqlen; };
    if (span(p.u_batch_stride, p.u_d_stride) >= lim || span(p.delta_batch_stride, p.delta_d_stride) >= lim ||
        span(p.out_batch_stride, p.out_d_stride) >= lim || span(p.z_batch_stride, p.z_d_stride) >= lim ||
        span(p.out_z_batch_stride, p.out_z_d_stride) >= lim || (int64_t)(p.dstate - 1) * p.B_dstate_stride + p.seqlen >= lim ||
        (int64_t)(p.dstate - 1) * p.C_dstate_stride + p.seqlen >= lim)
        return false;
    return true;
}

// reverse_from served by ONE launch: whole-vector rows whose workgroups share a B / C group (the LDS kernel), no sequence split
bool scan_fwd_pair_native_mixed(const vms_scan_fwd_params& p) {
    if (!(p.reverse_from > 0 && p.reverse_from < p.batch) || p.reverse) return false;
    vms_scan_fwd_params t = p;
    t.reverse_from = 0;
    if (scan_fwd_sg_wanted(t)) return true;   // one row per workgroup: no group condition
    if (p.seqlen % kPK != 0 || (p.dim / p.n_groups) % kLW != 0 || p.x_has_sub == 2) return false;
    return scan_fwd_pair_segments(t) <= 1;
}

template <typename T>
static int launch_pair(const vms_scan_fwd_params& p, hipStream_t stream) {
    const int tiles = (p.dim + kPRows - 1) / kPRows;
    dim3 grid(p.batch * tiles), block(kPRows * kWave);
    const bool rag = p.seqlen % kPK != 0;
    if (scan_fwd_sg_wanted(p)) {   // few rows, long sequences: the states of a row over the waves of a workgroup, one pass
        const dim3 grid_s(p.batch * p.dim), block_s(kSW * kWave);
        const size_t smem_s = sizeof(float) * kSgFloats;
        const bool mixed_s = p.reverse_from > 0 && p.reverse_from < p.batch;
#define VMS_LS(Z_, R_) hipLaunchKernelGGL((scan_fwd_sg_kernel<T, Z_, R_>), grid_s, block_s, smem_s, stream, p)
        if (mixed_s) { if (p.z) VMS_LS(true, 2); else VMS_LS(false, 2); }
        else if (p.reverse) { if (p.z) VMS_LS(true, 1); else VMS_LS(false, 1); }
        else { if (p.z) VMS_LS(true, 0); else VMS_LS(false, 0); }
#undef VMS_LS
        VMS_LAUNCH_CHECK();
        set_last_kernel(mixed_s ? "scan_fwd_sg+mixed" : "scan_fwd_sg");
        return VMS_OK;
    }
    int n_seg = 1;
    float2* carry = nullptr;
    if (p.workspace != nullptr && p.workspace_bytes >= scan_fwd_pair_ws_bytes(p) && p.x_has_sub != 2) {
        n_seg = scan_fwd_pair_segments(p);
        carry = static_cast<float2*>(p.workspace);
    }
    if (n_seg > 1) {
        dim3 cgrid(p.batch * tiles * (n_seg - 1));
        if (p.reverse) hipLaunchKernelGGL((scan_fwd_carry_kernel<T, true>), cgrid, block, 0, stream, p, n_seg, carry);
        else hipLaunchKernelGGL((scan_fwd_carry_kernel<T, false>), cgrid, block, 0, stream, p, n_seg, carry);
        grid = dim3(p.batch * tiles * n_seg);
    }
#ifdef VMS_FWD_NO_LDS
    constexpr bool lds_ok = false;            // A/B builds: the per-wave B / C kernel for every problem
#else
    const bool lds_ok = !rag && (p.dim / p.n_groups) % kLW == 0;   // a workgroup's rows share one B / C group
#endif
    const bool mixed = p.reverse_from > 0 && p.reverse_from < p.batch;
    // 64 KB of fp32 B / C + (x_has_sub == 3) 16 KB of parked checkpoints: two workgroups fill a CU's 160 KB exactly
#ifndef VMS_FWD_LDS_PAD
#define VMS_FWD_LDS_PAD 0   /* A/B builds (tools/variant.sh): bytes of unused LDS per workgroup = fewer resident waves per SIMD */
#endif
    const size_t smem_l = sizeof(float) * 2 * kLGroupFloats + (p.x_has_sub == 3 ? sizeof(float) * 2 * kLG * kWave * kLW : 0) + VMS_FWD_LDS_PAD;
    if (smem_l > 64 * 1024) {
        static PerDeviceOnce attr_once;
        const hipError_t arc = attr_once.run([&]() -> hipError_t {
            hipError_t e = hipSuccess;
#define VMS_ALN(Z_, R_, N_)                                                                                             \
            if (e == hipSuccess)                                                                                        \
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(&scan_fwd_lds_kernel<T, Z_, R_, true, false, N_>), \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024 + VMS_FWD_LDS_PAD);       \
            if (e == hipSuccess)                                                                                        \
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(&scan_fwd_lds_kernel<T, Z_, R_, true, true, N_>),  \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024 + VMS_FWD_LDS_PAD)
#define VMS_AL(Z_, R_) VMS_ALN(Z_, R_, kPN); VMS_ALN(Z_, R_, 0)
            VMS_AL(true, 0); VMS_AL(true, 1); VMS_AL(true, 2); VMS_AL(false, 0); VMS_AL(false, 1); VMS_AL(false, 2);
#undef VMS_AL
#undef VMS_ALN
            return e;
        });
        if (arc != hipSuccess) {
            set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d) failed: %s", 80 * 1024, hipGetErrorString(arc));
            return VMS_ERR_LAUNCH;
        }
    }
    const dim3 grid_l(p.batch * ((p.dim + kLW - 1) / kLW) * n_seg), block_l(kLW * kWave);
    // a row whose last 1024-element chunk holds <= 256 elements: the instantiation with the four-states-at-a-time tail form
#ifndef VMS_FWD_TAIL_QUAD
#define VMS_FWD_TAIL_QUAD 1   /* 0 (A/B builds, tools/variant.sh): every chunk as a whole-wave pass */
#endif
    const int tail_len = p.seqlen % (kWave * kPK);
    const bool tq = VMS_FWD_TAIL_QUAD && tail_len > 0 && tail_len <= 16 * kPK;
#define VMS_LLN(Z_, R_, S_, N_)                                                                                    \
    do {                                                                                                           \
        if (p.x_has_sub == 3) {                                                                                    \
            if (tq) hipLaunchKernelGGL((scan_fwd_lds_kernel<T, Z_, R_, true, true, N_>), grid_l, block_l, smem_l, stream, p, S_, carry);   \
            else hipLaunchKernelGGL((scan_fwd_lds_kernel<T, Z_, R_, true, false, N_>), grid_l, block_l, smem_l, stream, p, S_, carry);      \
        } else {                                                                                                   \
            if (tq) hipLaunchKernelGGL((scan_fwd_lds_kernel<T, Z_, R_, false, true, N_>), grid_l, block_l, smem_l, stream, p, S_, carry);  \
            else hipLaunchKernelGGL((scan_fwd_lds_kernel<T, Z_, R_, false, false, N_>), grid_l, block_l, smem_l, stream, p, S_, carry);     \
        }                                                                                                          \
    } while (0)
#define VMS_LL(Z_, R_, S_)                                                                                         \
    do {                                                                                                           \
        if (p.dstate == kPN) VMS_LLN(Z_, R_, S_, kPN);                                                             \
        else VMS_LLN(Z_, R_, S_, 0);                                                                               \
    } while (0)
#define VMS_L(Z_, R_)                                                                                              \
    do {                                                                                                           \
        if (rag) hipLaunchKernelGGL((scan_fwd_pair_kernel<T, Z_, R_, true>), grid, block, 0, stream, p, 1, carry); \
        else if (lds_ok) VMS_LL(Z_, R_ ? 1 : 0, n_seg);                                                            \
        else hipLaunchKernelGGL((scan_fwd_pair_kernel<T, Z_, R_, false>), grid, block, 0, stream, p, n_seg, carry); \
    } while (0)
    if (mixed) {   // reverse_from: only the LDS kernel, unsplit (scan_fwd_pair_native_mixed); everything else is split by the host
        if (p.z) VMS_LL(true, 2, 1);
        else VMS_LL(false, 2, 1);
    } else if (p.reverse) { if (p.z) VMS_L(true, true); else VMS_L(false, true); }
    else { if (p.z) VMS_L(true, false); else VMS_L(false, false); }
#undef VMS_L
#undef VMS_LL
#undef VMS_LLN
    VMS_LAUNCH_CHECK();
    // distinct names per kernel: a shape that silently falls off the LDS kernel must be visible to callers and tests
    set_last_kernel(mixed ? (p.dstate == kPN ? "scan_fwd_pair_lds+mixed" : "scan_fwd_pair_lds_n+mixed") : rag ? "scan_fwd_pair_ragged"
                        : lds_ok ? (n_seg > 1 ? "scan_fwd_pair_lds+split" : p.dstate == kPN ? "scan_fwd_pair_lds" : "scan_fwd_pair_lds_n")
                                 : (n_seg > 1 ? "scan_fwd_pair+split" : "scan_fwd_pair"));
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
