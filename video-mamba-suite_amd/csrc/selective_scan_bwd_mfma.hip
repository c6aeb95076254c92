// selective_scan_bwd_mfma.hip -- the fast backward selective scan for gfx950 (wave64).
//
// Same math as selective_scan_bwd.hip (which stays as the generic kernel); this one is taken
// when B and C are input dependent, dstate == 16, I/O is 16-byte aligned and the forward left
// 128-element sub-checkpoints in x (vms_hip.h).  Replaces selective_scan_bwd_kernel
// (mamba/csrc/selective_scan/selective_scan_bwd_kernel.cuh:75-489).
//
// Decomposition (DESIGN.md "scan backward, fast path"):
//   * a WAVE owns 4 rows: lane = 16*r + j, row r = lane>>4 (a DPP "row"), j = lane&15 owns the K
//     consecutive elements [K*j, K*j+K) of the row's current 16*K-element chunk.  Both scans
//     (forward re-scan, adjoint suffix scan) are 4-step DPP row scans; nothing crosses a DPP row.
//   * the reduction of dB / dC over rows -- 134M fp32 atomics with 1024-way contention in the
//     reference (bwd_kernel.cuh:297-316) -- is done by the matrix pipe, which is otherwise idle:
//       D(16x16) += A(16x4) * B(4x16),  v_mfma_f32_16x16x4_f32, exact fp32
//     with B[k][j] = the lane's value (k = row r, j = lane column: exactly the lane layout) and
//     A[i][k] = (i == e) a 0/1 selector: the sum over the 4 rows of element e lands in row e of D.
//     After K such MFMAs, lane (q, j) holds in its 4 accumulator registers the 4-row sums of the
//     4 consecutive positions K*j + 4q + {0..3}: one 16-byte vector per lane, no shuffles.
//   * the W waves (4W rows) of a workgroup then combine their vectors through LDS (plain b128
//     writes / reads -- ds_add_f32 runs at 0.33 lanes/clk/CU on this chip) once
//     per 4 states, and one fp32 global atomic per 4W rows goes to dB / dC.
//   * per-(row, state) carries (adjoint entering from the right, a of the next chunk's first
//     element, dA accumulator) live in ONE register each: lane j of a row keeps the value of
//     state n = j and hands it out with a DPP row broadcast.
#include <stdlib.h>

#include <type_traits>

#include "vms_common.cuh"

namespace vms {

#ifndef VMS_BWD_UNROLL
#define VMS_BWD_UNROLL 1  // states in flight per wave (ILP: only 2 waves per SIMD exist at config 2)
#endif
constexpr int kMN = 16;   // dstate handled by this kernel
constexpr int kMSG = 4;   // states between two cross-wave reductions

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;

// compile-time loop: the state index selects DPP controls, which must be immediates
template <int I, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, E>(f);
    }
}

// K elements as raw 16-byte vectors (issued early, converted where they are consumed)
template <typename T, int K>
struct RawVec {
    static constexpr int EPV = 16 / sizeof(T);
    vec_t<T, EPV> v[K / EPV];
    // the fast path only runs with seqlen % K == 0, so a lane's K elements are all valid or all
    // past the end: invalid lanes read the (always valid) start of the row and are zeroed -- no branch
    __device__ __forceinline__ void load(const T* __restrict__ base, uint32_t off, bool valid) {
        const vec_t<T, EPV>* vp = reinterpret_cast<const vec_t<T, EPV>*>(base + (valid ? off : 0u));
#pragma unroll
        for (int i = 0; i < K / EPV; ++i) {
            v[i] = vp[i];
            if (!valid) v[i] = vec_t<T, EPV>{};
        }
    }
    __device__ __forceinline__ void widen(float (&out)[K]) const {
#pragma unroll
        for (int i = 0; i < K; ++i) out[i] = static_cast<float>(v[i / EPV][i % EPV]);
    }
};

template <int CTRL>
__device__ __forceinline__ float rdpp(float old, float src) {
    return dpp_mov<CTRL, 0xf>(old, src);
}
constexpr int DPP_ROW_NEWBCAST0 = 0x150;
constexpr int DPP_ROW_ROR0 = 0x120;

// inclusive scans of the monoid (a, x) inside each 16-lane row
__device__ __forceinline__ void row_scan(float& a, float& x) {
#define VMS_S(C) { float xp = rdpp<C>(0.f, x), ap = rdpp<C>(1.f, a); x = fmaf(a, xp, x); a *= ap; }
    VMS_S(DPP_ROW_SHR1) VMS_S(DPP_ROW_SHR2) VMS_S(DPP_ROW_SHR4) VMS_S(DPP_ROW_SHR8)
#undef VMS_S
}
__device__ __forceinline__ void row_scan_rev(float& a, float& x) {
#define VMS_S(C) { float xp = rdpp<C>(0.f, x), ap = rdpp<C>(1.f, a); x = fmaf(a, xp, x); a *= ap; }
    VMS_S(DPP_ROW_SHL1) VMS_S(DPP_ROW_SHL2) VMS_S(DPP_ROW_SHL4) VMS_S(DPP_ROW_SHL8)
#undef VMS_S
}
// value of lane (row, n) for a run-time n: byte_index = ((lane & 48) | n) * 4
__device__ __forceinline__ float row_bcast(float v, int byte_index) {
#ifndef VMS_ABL_NOBCAST
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_index, __builtin_bit_cast(int, v)));
#else
    return v + byte_index;
#endif
}
__device__ __forceinline__ float row_allsum(float v) {
    v += rdpp<DPP_ROW_ROR0 + 1>(0.f, v);
    v += rdpp<DPP_ROW_ROR0 + 2>(0.f, v);
    v += rdpp<DPP_ROW_ROR0 + 4>(0.f, v);
    v += rdpp<DPP_ROW_ROR0 + 8>(0.f, v);
    return v;
}

template <typename T, int K, int W, bool HZ>
__global__ __launch_bounds__(W* kWave) void scan_bwd_mfma_kernel(const vms_scan_bwd_params q, const int dbg) {
    const vms_scan_fwd_params& p = q.f;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    lds_f32x4* slab = (lds_f32x4*)smem;  // [wave][state%4][tensor][lane < 4K]
    constexpr int N = kMN;
    constexpr int CH = 16 * K;        // elements per row per iteration
    constexpr int SL = 4 * K;         // lanes of a wave holding reduced vectors (D rows 0..K-1)
    constexpr int kRows = 4 * W;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, r = lane >> 4;
    // consecutive workgroups share a row tile across batches -> batch = blockIdx % batch keeps the
    // B/C of one batch on one XCD's L2 when batch == 8
    const int b = blockIdx.x % p.batch;
    const int d0 = (blockIdx.x / p.batch) * kRows;
    const int d = d0 + wave * 4 + r;
    const bool row_ok = d < p.dim;
    const int dc = row_ok ? d : p.dim - 1;
    const int g = d0 / (p.dim / p.n_groups);  // host guarantees one group per workgroup
    const int L = p.seqlen;

    // uniform base pointers (SGPRs) + one 32-bit element offset per lane and tensor (the host
    // guarantees every tensor spans < 2^31 elements on this path): keeps the row addressing out of
    // 64-bit VGPR pairs
    const T* const u_b = static_cast<const T*>(p.u);
    const T* const dt_b = static_cast<const T*>(p.delta);
    const T* const dout_b = static_cast<const T*>(q.dout);
    T* const du_b = static_cast<T*>(q.du);
    T* const ddelta_b = static_cast<T*>(q.ddelta);
    const T* const z_b = static_cast<const T*>(p.z);
    const T* const outp_b = static_cast<const T*>(p.out);
    T* const dz_b = static_cast<T*>(q.dz);
    T* const out_z_b = static_cast<T*>(p.out_z);
#define VMS_OFF(bs, ds) static_cast<uint32_t>((int64_t)b * (bs) + (int64_t)dc * (ds))
    const uint32_t o_u = VMS_OFF(p.u_batch_stride, p.u_d_stride);
    const uint32_t o_dt = VMS_OFF(p.delta_batch_stride, p.delta_d_stride);
    const uint32_t o_dout = VMS_OFF(q.dout_batch_stride, q.dout_d_stride);
    const uint32_t o_du = VMS_OFF(q.du_batch_stride, q.du_d_stride);
    const uint32_t o_ddelta = VMS_OFF(q.ddelta_batch_stride, q.ddelta_d_stride);
    const uint32_t o_z = HZ ? VMS_OFF(p.z_batch_stride, p.z_d_stride) : 0u;
    const uint32_t o_out = HZ ? VMS_OFF(p.out_batch_stride, p.out_d_stride) : 0u;
    const uint32_t o_dz = HZ ? VMS_OFF(q.dz_batch_stride, q.dz_d_stride) : 0u;
    const uint32_t o_outz = (HZ && p.out_z) ? VMS_OFF(p.out_z_batch_stride, p.out_z_d_stride) : 0u;
#undef VMS_OFF
    const T* Bv = static_cast<const T*>(p.B) + (int64_t)b * p.B_batch_stride + (int64_t)g * p.B_group_stride;
    const T* Cv = static_cast<const T*>(p.C) + (int64_t)b * p.C_batch_stride + (int64_t)g * p.C_group_stride;
    float* dBg = q.dB + (int64_t)b * q.dB_batch_stride + (int64_t)g * q.dB_group_stride;
    float* dCg = q.dC + (int64_t)b * q.dC_batch_stride + (int64_t)g * q.dC_group_stride;
    const uint32_t o_x = static_cast<uint32_t>(((int64_t)b * p.dim + dc) * p.n_chunks * p.x_chunk_stride);
    const float* const x_b = static_cast<const float*>(p.x);
    const float Dd = p.D ? static_cast<const float*>(p.D)[dc] : 0.f;
    const float bias = p.delta_bias ? static_cast<const float*>(p.delta_bias)[dc] : 0.f;
    // A of the lane's row: lane j keeps A[d][j]; handed out per state by a row broadcast
    const float A_mine = static_cast<const float*>(p.A)[(int64_t)dc * p.A_d_stride + (int64_t)j * p.A_dstate_stride];

    float sel[K];  // MFMA row selectors
#pragma unroll
    for (int e = 0; e < K; ++e) sel[e] = j == e ? 1.f : 0.f;

    float gcar = 0.f;   // adjoint entering this chunk from the right, state n = j
    float anx = 1.f;    // a of the first element of the chunk to the right, state n = j
    float dAacc = 0.f;  // dA[d][j]
    float dD_acc = 0.f, dbias_acc = 0.f;

    const int n_c = (L + CH - 1) / CH;
    for (int c = n_c - 1; c >= 0; --c) {
        const int l0 = c * CH + j * K;
        const int nv = row_ok ? L - l0 : 0;
        const int nvb = L - l0;  // B/C validity does not depend on the row
        const bool okb = nvb > 0, ok = nv > 0;
        RawVec<T, K> rawB, rawC;
        // B / C: uniform per-chunk base (cannot be hoisted out of the chunk loop) + lane offset
        const T* const Bc = Bv + c * CH;
        const T* const Cc = Cv + c * CH;
        const uint32_t jo = j * K;
        rawB.load(Bc, jo, okb);
        rawC.load(Cc, jo, okb);
        float uv[K], dl[K], dy[K], duv[K], ddl[K];
        {
            RawVec<T, K> t0, t1, t2;
            t0.load(u_b, o_u + l0, ok);
            t1.load(dt_b, o_dt + l0, ok);
            t2.load(dout_b, o_dout + l0, ok);
            t0.widen(uv);
            t1.widen(dl);
            t2.widen(dy);
        }
        // state entering the chunk = 128-element sub-checkpoint (vms_hip.h); lane j loads state j
        float hck = 0.f;
        if (c > 0) {
            const int e128 = c * (CH / 128) - 1;  // index of the sub-checkpoint ending at c*CH
            hck = x_b[o_x + (uint32_t)((e128 >> 4) * (int)p.x_chunk_stride + 2 * N + (e128 & 15) * N + j)];
        }
#pragma unroll
        for (int i = 0; i < K; ++i) {
            float t = dl[i] + bias;
            if (p.delta_softplus) t = softplusf_(t);
            dl[i] = ok ? t : 0.f;
        }
        if (HZ) {
            float zv[K], ov[K], dzv[K];
            {
                RawVec<T, K> t0, t1;
                t0.load(z_b, o_z + l0, ok);
                t1.load(outp_b, o_out + l0, ok);
                t0.widen(zv);
                t1.widen(ov);
            }
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const float s = sigmoidf_(zv[i]);
                const float silu = zv[i] * s;
                dzv[i] = dy[i] * ov[i] * s * (1.f + zv[i] * (1.f - s));
                dy[i] *= silu;
                ov[i] *= silu;
            }
            if (ok) {
                store_blocked<T, K, true>(dz_b + (o_dz + l0), K, dzv);
                if (out_z_b) store_blocked<T, K, true>(out_z_b + (o_outz + l0), K, ov);
            }
        }
#pragma unroll
        for (int i = 0; i < K; ++i) {
            duv[i] = Dd * dy[i];
            dD_acc = fmaf(dy[i], uv[i], dD_acc);
            ddl[i] = 0.f;
        }
        // rolled on purpose: the fully unrolled 16-state body does not fit the instruction cache
        // (measured 11 ms vs < 1 ms); the per-state row broadcasts therefore use ds_bpermute
        // (run-time lane index) instead of DPP row_newbcast (immediate lane index)
#pragma unroll VMS_BWD_UNROLL
        for (int n = 0; n < N; ++n) {
            const int bsrc = ((lane & 48) | n) << 2;  // byte index of lane n of this lane's row
            const float Araw = row_bcast(A_mine, bsrc);
            const float An = Araw * kLog2e;
            float Bn[K], Cn[K];
            rawB.widen(Bn);
            rawC.widen(Cn);
#ifdef VMS_ABL_NOBCLOAD
            if (n + 1 < N && dbg == 12345) {
#else
            if (n + 1 < N) {  // prefetch the next state's B/C while this one computes
#endif
                rawB.load(Bc + (int64_t)(n + 1) * p.B_dstate_stride, jo, okb);
                rawC.load(Cc + (int64_t)(n + 1) * p.C_dstate_stride, jo, okb);
            }
            // ---- forward re-scan ----
            float a[K], xs[K];
            float pa = 1.f, px = 0.f;
#pragma unroll
            for (int i = 0; i < K; ++i) {
#ifndef VMS_ABL_NOEXP
                a[i] = fast_exp2(dl[i] * An);
#else
                a[i] = dl[i] * An + 1.f;
#endif
                xs[i] = dl[i] * uv[i] * Bn[i];  // b_i for now
                px = fmaf(a[i], px, xs[i]);
                pa *= a[i];
            }
#ifndef VMS_ABL_NOSCAN
            row_scan(pa, px);
#endif
            const float ea = rdpp<DPP_ROW_SHR1>(1.f, pa);
            const float ex = rdpp<DPP_ROW_SHR1>(0.f, px);
            const float hin = row_bcast(hck, bsrc);
            const float xseed = fmaf(ea, hin, ex);  // state entering this lane's first element
            float xrun = xseed;
            // ---- adjoint: g_i = C_i dy_i + a_{i+1} g_{i+1} ----
            const float anx_n = row_bcast(anx, bsrc);
            const float a_right = rdpp<DPP_ROW_SHL1>(anx_n, a[0]);  // lane 15 of the row <- next chunk
            float ra = 1.f, rg = 0.f;
#pragma unroll
            for (int i = K - 1; i >= 0; --i) {
                const float alpha = i == K - 1 ? a_right : a[i + 1];
                rg = fmaf(alpha, rg, dy[i] * Cn[i]);
                ra *= alpha;
            }
#ifndef VMS_ABL_NOSCAN
            row_scan_rev(ra, rg);
#endif
            const float esa = rdpp<DPP_ROW_SHL1>(1.f, ra);
            const float esx = rdpp<DPP_ROW_SHL1>(0.f, rg);
            const float gin = row_bcast(gcar, bsrc);
            float grun = fmaf(esa, gin, esx);
            // new carries = values at the row's lane 0
            const float gout = rdpp<DPP_ROW_NEWBCAST0 + 0>(0.f, fmaf(ra, gin, rg));
            const float afirst = rdpp<DPP_ROW_NEWBCAST0 + 0>(0.f, a[0]);
            if (j == n) { gcar = gout; anx = afirst; }
            // forward pass B: x_i (xs holds b_i on entry)
#pragma unroll
            for (int i = 0; i < K; ++i) {
                xrun = fmaf(a[i], xrun, xs[i]);
                xs[i] = xrun;
            }
            // adjoint pass B + all per-element outputs, right to left
            f32x4 accB = {0.f, 0.f, 0.f, 0.f}, accC = {0.f, 0.f, 0.f, 0.f};
            float dA_loc = 0.f;
#pragma unroll
            for (int i = K - 1; i >= 0; --i) {
                const float alpha = i == K - 1 ? a_right : a[i + 1];
                grun = fmaf(alpha, grun, dy[i] * Cn[i]);
                const float gx = grun;
                const float ddelta_u = gx * Bn[i];
                const float gax = gx * (a[i] * (i == 0 ? xseed : xs[i - 1]));  // g * a_i x_{i-1}
                duv[i] = fmaf(ddelta_u, dl[i], duv[i]);
                ddl[i] = fmaf(ddelta_u, uv[i], ddl[i]);
                ddl[i] = fmaf(Araw, gax, ddl[i]);
                dA_loc = fmaf(dl[i], gax, dA_loc);
                const float dBi = gx * (dl[i] * uv[i]);
                const float dCi = dy[i] * xs[i];
#ifndef VMS_ABL_NOMFMA
                accB = __builtin_amdgcn_mfma_f32_16x16x4f32(sel[i], dBi, accB, 0, 0, 0);
                accC = __builtin_amdgcn_mfma_f32_16x16x4f32(sel[i], dCi, accC, 0, 0, 0);
#else
                accB[i & 3] += dBi * sel[i];
                accC[i & 3] += dCi * sel[i];
#endif
            }
            const float dA_tot = row_allsum(dA_loc);
            if (j == n) dAacc += dA_tot;
            // 4-row sums of this state -> LDS (only D rows 0..K-1, i.e. lanes < 4K, carry data)
            if (lane < SL) {
                slab[((wave * kMSG + (n % kMSG)) * 2 + 0) * SL + lane] = accB;
                slab[((wave * kMSG + (n % kMSG)) * 2 + 1) * SL + lane] = accC;
            }
            if (n % kMSG == kMSG - 1 && !(dbg & 4)) {
                __syncthreads();
                // (tensor, state, slab lane) -> sum over the W waves, 4 atomics per thread
                for (int t = threadIdx.x; t < 2 * kMSG * SL; t += W * kWave) {
                    const int ten = t / (kMSG * SL), st = (t / SL) % kMSG, pl = t % SL;
                    f32x4 s = slab[((0 * kMSG + st) * 2 + ten) * SL + pl];
#pragma unroll
                    for (int w = 1; w < W; ++w) s += slab[((w * kMSG + st) * 2 + ten) * SL + pl];
                    const int nn = n - (kMSG - 1) + st;
                    const int lo = c * CH + (pl & 15) * K + 4 * (pl >> 4);  // first of 4 positions
                    float* dst = (ten == 0 ? dBg + (int64_t)nn * q.dB_dstate_stride
                                           : dCg + (int64_t)nn * q.dC_dstate_stride) + lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (lo + e < L && !(dbg & 1)) atomicAdd(dst + e, s[e]);
                }
                __syncthreads();
            }
        }
        {
            float raw[K];
            RawVec<T, K> t0;
            t0.load(dt_b, o_dt + l0, ok);
            t0.widen(raw);
#pragma unroll
            for (int i = 0; i < K; ++i) {
                if (p.delta_softplus) {
                    const float rr = raw[i] + bias;
                    ddl[i] = rr <= 20.f ? ddl[i] * sigmoidf_(rr) : ddl[i];
                }
                dbias_acc += ok ? ddl[i] : 0.f;
            }
        }
        if (ok) {
            store_blocked<T, K, true>(du_b + (o_du + l0), K, duv);
            store_blocked<T, K, true>(ddelta_b + (o_ddelta + l0), K, ddl);
        }
    }
    const float dD_tot = row_allsum(dD_acc), db_tot = row_allsum(dbias_acc);
    if (row_ok) {
        if (q.dD && j == 0) atomicAdd(q.dD + d, dD_tot);
        if (q.ddelta_bias && j == 0) atomicAdd(q.ddelta_bias + d, db_tot);
        atomicAdd(q.dA + (int64_t)d * q.dA_d_stride + (int64_t)j * q.dA_dstate_stride, dAacc);
    }
}

#ifndef VMS_BWD_K
#define VMS_BWD_K 8   // elements per lane
#endif
#ifndef VMS_BWD_W
#define VMS_BWD_W 8   // waves per workgroup: 4*W rows share one set of dB/dC atomics
#endif
constexpr int kMK = VMS_BWD_K;
constexpr int kMW = VMS_BWD_W;

bool scan_bwd_mfma_eligible(const vms_scan_bwd_params& q, bool vec) {
    const vms_scan_fwd_params& p = q.f;
    if (!vec || !p.is_variable_B || !p.is_variable_C || p.dstate != kMN || !p.x || !p.x_has_sub) return false;
    const int dpg = p.dim / p.n_groups;
    if (dpg % (4 * kMW) != 0) return false;  // a workgroup's rows must share one B/C group
    if (p.seqlen % kMK != 0) return false;   // a lane's K elements are all in range or all out
    // 32-bit element offsets inside the kernel
    const int64_t lim = (int64_t)1 << 31;
    auto span = [&](int64_t bs, int64_t ds) { return (p.batch - 1) * bs + (p.dim - 1) * ds + p.seqlen; };
    if (span(p.u_batch_stride, p.u_d_stride) >= lim || span(p.delta_batch_stride, p.delta_d_stride) >= lim ||
        span(q.dout_batch_stride, q.dout_d_stride) >= lim || span(q.du_batch_stride, q.du_d_stride) >= lim ||
        span(q.ddelta_batch_stride, q.ddelta_d_stride) >= lim || span(p.z_batch_stride, p.z_d_stride) >= lim ||
        span(p.out_batch_stride, p.out_d_stride) >= lim || span(q.dz_batch_stride, q.dz_d_stride) >= lim ||
        span(p.out_z_batch_stride, p.out_z_d_stride) >= lim ||
        (int64_t)p.batch * p.dim * p.n_chunks * p.x_chunk_stride >= lim)
        return false;
    return true;
}

template <typename T, int K, int W>
static int launch_mfma(const vms_scan_bwd_params& q, hipStream_t stream) {
    const vms_scan_fwd_params& p = q.f;
    const int tiles = (p.dim + 4 * W - 1) / (4 * W);
    dim3 grid(p.batch * tiles), block(W * kWave);
    const size_t smem = sizeof(float) * 4 * (4 * K) * 2 * kMSG * W;
    const int dbg = getenv("VMS_DEBUG") ? atoi(getenv("VMS_DEBUG")) : 0;  // profiling knob
    if (p.z) hipLaunchKernelGGL((scan_bwd_mfma_kernel<T, K, W, true>), grid, block, smem, stream, q, dbg);
    else hipLaunchKernelGGL((scan_bwd_mfma_kernel<T, K, W, false>), grid, block, smem, stream, q, dbg);
    VMS_LAUNCH_CHECK();
    return VMS_OK;
}

int launch_scan_bwd_mfma(const vms_scan_bwd_params& q, hipStream_t stream) {
    switch (q.f.dtype) {
        case VMS_BF16: return launch_mfma<bf16_t, kMK, kMW>(q, stream);
        case VMS_F16: return launch_mfma<f16_t, kMK, kMW>(q, stream);
        default: return launch_mfma<float, kMK, kMW>(q, stream);
    }
}

}  // namespace vms
