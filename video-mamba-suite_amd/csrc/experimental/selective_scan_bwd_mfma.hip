// selective_scan_bwd_mfma.hip -- the fast backward selective scan for gfx950 (wave64).
//
// Same math as selective_scan_bwd.hip (which stays as the generic kernel); this one is taken
// when B and C are input dependent, dstate == 16, I/O is 16-byte aligned, seqlen % 8 == 0 and
// the forward left 128-element sub-checkpoints in x (vms_hip.h).  Replaces
// selective_scan_bwd_kernel (mamba/csrc/selective_scan/selective_scan_bwd_kernel.cuh:75-489).
//
// Decomposition (DESIGN.md "scan backward, fast path"):
//   * a WAVE owns 4 rows x 8 of the 16 states: lane = 16*r + j, row r = lane>>4 (a DPP "row"),
//     j = lane&15 owns the 8 consecutive elements [8j, 8j+8) of the row's current 128-element
//     chunk.  Both scans (forward re-scan, adjoint suffix scan) are 4-step DPP row scans issued
//     as fused v_fmac_f32_dpp / v_mul_f32_dpp pairs; nothing crosses a DPP row.
//   * a workgroup = 16 waves = 8 row quads x 2 state halves = 32 rows.  The state split exists
//     for occupancy: one wave can issue a VALU instruction only every ~8.5 cycles on this chip
//     (tools/microbench/microbench.hip), so the kernel needs 4 waves per SIMD, and (8, 8192, 1024, 16) has
//     only 2048 row quads for 1024 SIMDs.
//   * the reduction of dB / dC over rows -- 134M fp32 atomics with 1024-way contention in the
//     reference (bwd_kernel.cuh:297-316) -- is done by the matrix pipe, which is otherwise idle:
//       D(16x16) += A(16x4) * B(4x16),  v_mfma_f32_16x16x4_f32, exact fp32
//     with B[k][j] = the lane's value (k = row r, j = lane column: exactly the lane layout) and
//     A[i][k] = (i == e) a 0/1 selector: the sum over the 4 rows of element e lands in row e of D.
//     After 8 such MFMAs, lane (q < 2, j) holds in its 4 accumulator registers the 4-row sums of
//     the 4 consecutive positions 8j + 4q + {0..3}: one 16-byte vector per lane, no shuffles.
//   * the 8 row quads of a workgroup then combine their vectors through LDS (plain b128 writes /
//     reads -- ds_add_f32 runs at 0.33 lanes/clk/CU on this chip) once per 4 states, and one fp32
//     global atomic per 32 rows goes to dB / dC.  Barriers order LDS traffic only (no vmcnt drain).
//   * du / ddelta are sums over all 16 states: the upper state half hands its partial sums to the
//     lower half through LDS once per chunk.
//   * per-(row, state) carries (adjoint entering from the right, a of the next chunk's first
//     element, dA accumulator) live in ONE register each: lane j of a row keeps the value of
//     state n = j and hands it out with ds_bpermute (prefetched one state ahead).
#include "vms_common.h"

namespace vms {

constexpr int kMN = 16;   // dstate handled by this kernel
#ifndef VMS_BWD_K
#define VMS_BWD_K 8
#endif
constexpr int kMK = VMS_BWD_K;  // elements per lane (8 or 16)
constexpr int kMQ = 8;    // row quads per workgroup
constexpr int kMRows = 4 * kMQ;
constexpr int kMSG = 4;   // states between two cross-quad reductions
#ifndef VMS_BWD_NSPLIT
#define VMS_BWD_NSPLIT 1  // waves sharing a row quad, each taking 16 / NSPLIT states
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;

// K elements as raw 16-byte vectors (issued early, converted where they are consumed)
template <typename T, int K>
struct RawVec {
    static constexpr int EPV = 16 / sizeof(T);
    vec_t<T, EPV> v[K / EPV];
    // seqlen % K == 0 on this path, so a lane's K elements are all valid or all past the end:
    // invalid lanes read the (always valid) start of the buffer and are zeroed -- no branch
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

// Forward inclusive scan of (pa, px) and suffix inclusive scan of (ra, rg) inside each 16-lane
// row, interleaved: one DPP-fused VOP2 per monoid component and step (x += dpp(x) * a ;
// a *= dpp(a); lanes whose DPP source falls outside the row are not written = identity).  The
// interleaving also provides the 2 wait states a DPP read needs after a VALU write of its source.
__device__ __forceinline__ void row_scan_pair(float& pa, float& px, float& ra, float& rg) {
#define VMS_STEP(S)                                                                   \
    "v_fmac_f32_dpp %0, %0, %1 row_shr:" #S " row_mask:0xf bank_mask:0xf\n\t"          \
    "v_fmac_f32_dpp %2, %2, %3 row_shl:" #S " row_mask:0xf bank_mask:0xf\n\t"          \
    "v_mul_f32_dpp %1, %1, %1 row_shr:" #S " row_mask:0xf bank_mask:0xf\n\t"           \
    "v_mul_f32_dpp %3, %3, %3 row_shl:" #S " row_mask:0xf bank_mask:0xf\n\t"
    asm volatile("s_nop 1\n\t" VMS_STEP(1) VMS_STEP(2) VMS_STEP(4) VMS_STEP(8) "s_nop 1"
                 : "+v"(px), "+v"(pa), "+v"(rg), "+v"(ra));
#undef VMS_STEP
}

// workgroup barrier that orders LDS traffic only (no vmcnt drain)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// value of lane (row, n) for a run-time n: byte_index = ((lane & 48) | n) * 4
__device__ __forceinline__ float row_bcast(float v, int byte_index) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_index, __builtin_bit_cast(int, v)));
}
__device__ __forceinline__ float row_allsum(float v) {
    v += rdpp<DPP_ROW_ROR0 + 1>(0.f, v);
    v += rdpp<DPP_ROW_ROR0 + 2>(0.f, v);
    v += rdpp<DPP_ROW_ROR0 + 4>(0.f, v);
    v += rdpp<DPP_ROW_ROR0 + 8>(0.f, v);
    return v;
}

template <typename T, bool HZ, int NSPLIT>
__global__ __launch_bounds__(kMQ* NSPLIT* kWave) void scan_bwd_mfma_kernel(const vms_scan_bwd_params q) {
    constexpr int kMNS = kMN / NSPLIT;  // states per wave
    const vms_scan_fwd_params& p = q.f;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int K = kMK, N = kMN;
    constexpr int CH = 16 * K;   // elements per row per iteration (128)
    constexpr int SL = 4 * K;    // lanes of a wave holding reduced vectors (D rows 0..K-1)
    lds_f32x4* slab = (lds_f32x4*)smem;                           // [quad][half][state%4][tensor][SL]
    lds_f32x4* xchg = slab + kMQ * NSPLIT * kMSG * 2 * SL;         // [quad][4 vectors][64 lanes]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int quad = wave / NSPLIT, half = wave % NSPLIT;
    const int j = lane & 15, r = lane >> 4;
    // consecutive workgroups share a row tile across batches -> batch = blockIdx % batch keeps the
    // B/C of one batch on one XCD's L2 when batch == 8
    const int b = blockIdx.x % p.batch;
    const int d0 = (blockIdx.x / p.batch) * kMRows;
    const int d = d0 + quad * 4 + r;
    const bool row_ok = d < p.dim;
    const int dc = row_ok ? d : p.dim - 1;
    const int g = d0 / (p.dim / p.n_groups);  // host guarantees one group per workgroup
    const int L = p.seqlen;
    const int n0 = half * kMNS;

    // uniform base pointers (SGPRs) + 32-bit element offsets per lane, rebuilt where used (the
    // host guarantees every tensor spans < 2^31 elements on this path)
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
    const T* Bv = static_cast<const T*>(p.B) + (int64_t)b * p.B_batch_stride + (int64_t)g * p.B_group_stride;
    const T* Cv = static_cast<const T*>(p.C) + (int64_t)b * p.C_batch_stride + (int64_t)g * p.C_group_stride;
    float* dBg = q.dB + (int64_t)b * q.dB_batch_stride + (int64_t)g * q.dB_group_stride;
    float* dCg = q.dC + (int64_t)b * q.dC_batch_stride + (int64_t)g * q.dC_group_stride;
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
        const bool okb = l0 < L, ok = okb && row_ok;
        // B / C: uniform per-chunk base + lane offset; first state of this wave
        const T* const Bc = Bv + c * CH + (int64_t)n0 * p.B_dstate_stride;
        const T* const Cc = Cv + c * CH + (int64_t)n0 * p.C_dstate_stride;
        const uint32_t jo = j * K;
        RawVec<T, K> rawB, rawC;
        rawB.load(Bc, jo, okb);
        rawC.load(Cc, jo, okb);
        float uv[K], dl[K], dy[K], duv[K], ddl[K];
        {
            RawVec<T, K> t0, t1, t2;
            t0.load(u_b, VMS_OFF(p.u_batch_stride, p.u_d_stride) + l0, ok);
            t1.load(dt_b, VMS_OFF(p.delta_batch_stride, p.delta_d_stride) + l0, ok);
            t2.load(dout_b, VMS_OFF(q.dout_batch_stride, q.dout_d_stride) + l0, ok);
            t0.widen(uv);
            t1.widen(dl);
            t2.widen(dy);
        }
        // state entering the chunk = 128-element sub-checkpoint c-1 (vms_hip.h); lane j loads state j
        float hck = 0.f;
        if (c > 0) {
            const uint32_t o_x = static_cast<uint32_t>(((int64_t)b * p.dim + dc) * p.n_chunks * p.x_chunk_stride);
            const int e128 = c * (CH / 128) - 1;  // index of the 128-element sub-checkpoint ending at c*CH
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
                t0.load(z_b, VMS_OFF(p.z_batch_stride, p.z_d_stride) + l0, ok);
                t1.load(outp_b, VMS_OFF(p.out_batch_stride, p.out_d_stride) + l0, ok);
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
            if (ok && half == 0) {
                if (q.dz_accumulate) {  // dz += (vms_hip.h)
                    float old[K];
                    load_blocked<T, K, true>(dz_b + (VMS_OFF(q.dz_batch_stride, q.dz_d_stride) + l0), K, old);
#pragma unroll
                    for (int i = 0; i < K; ++i) dzv[i] += old[i];
                }
                store_blocked<T, K, true>(dz_b + (VMS_OFF(q.dz_batch_stride, q.dz_d_stride) + l0), K, dzv);
                if (out_z_b) store_blocked<T, K, true>(out_z_b + (VMS_OFF(p.out_z_batch_stride, p.out_z_d_stride) + l0), K, ov);
            }
        }
#pragma unroll
        for (int i = 0; i < K; ++i) {
            duv[i] = half == 0 ? Dd * dy[i] : 0.f;
            if (half == 0) dD_acc = fmaf(dy[i], uv[i], dD_acc);
            ddl[i] = 0.f;
        }
        // row broadcasts for the wave's first state (later states are prefetched inside the loop)
        float bc_A, bc_h, bc_anx, bc_g;
        {
            const int b0 = ((lane & 48) | n0) << 2;
            bc_A = row_bcast(A_mine, b0);
            bc_h = row_bcast(hck, b0);
            bc_anx = row_bcast(anx, b0);
            bc_g = row_bcast(gcar, b0);
        }
        // One state.  The B/C of the NEXT state are requested first, into the other register set:
        // explicit double buffering (two named sets, loop unrolled by exactly 2) -- with a single
        // loop-carried set the compiler rotates the loop and waits for the load right after issue.
        auto do_state = [&](const int ni, RawVec<T, K>& curB, RawVec<T, K>& curC, RawVec<T, K>& nxtB,
                            RawVec<T, K>& nxtC) __attribute__((always_inline)) {
            const int n = n0 + ni;
            if (ni + 1 < kMNS) {
                nxtB.load(Bc + (int64_t)(ni + 1) * p.B_dstate_stride, jo, okb);
                nxtC.load(Cc + (int64_t)(ni + 1) * p.C_dstate_stride, jo, okb);
            }
            const float Araw = bc_A, hin = bc_h, anx_n = bc_anx, gin = bc_g;
            {   // next state's broadcasts (lane n+1 is not touched by this state's carry updates)
                const int bnext = ((lane & 48) | ((n + 1) & (N - 1))) << 2;
                bc_A = row_bcast(A_mine, bnext);
                bc_h = row_bcast(hck, bnext);
                bc_anx = row_bcast(anx, bnext);
                bc_g = row_bcast(gcar, bnext);
            }
            const float An = Araw * kLog2e;
            float Bn[K], Cn[K];
            curB.widen(Bn);
            curC.widen(Cn);
            // ---- local scans: forward (a, b) and adjoint (alpha = a_{i+1}, c = C dy) ----
            float a[K], xs[K];
            float pa = 1.f, px = 0.f;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                a[i] = fast_exp2(dl[i] * An);
                xs[i] = dl[i] * uv[i] * Bn[i];  // b_i for now
                px = fmaf(a[i], px, xs[i]);
                pa *= a[i];
            }
            const float a_right = rdpp<DPP_ROW_SHL1>(anx_n, a[0]);  // lane 15 of the row <- next chunk
            float ra = 1.f, rg = 0.f;
#pragma unroll
            for (int i = K - 1; i >= 0; --i) {
                const float alpha = i == K - 1 ? a_right : a[i + 1];
                Cn[i] *= dy[i];  // c_i
                rg = fmaf(alpha, rg, Cn[i]);
                ra *= alpha;
            }
            row_scan_pair(pa, px, ra, rg);
            const float ea = rdpp<DPP_ROW_SHR1>(1.f, pa);
            const float ex = rdpp<DPP_ROW_SHR1>(0.f, px);
            const float xseed = fmaf(ea, hin, ex);  // state entering this lane's first element
            const float esa = rdpp<DPP_ROW_SHL1>(1.f, ra);
            const float esx = rdpp<DPP_ROW_SHL1>(0.f, rg);
            float grun = fmaf(esa, gin, esx);
            // new carries = values at the row's lane 0
            const float gout = rdpp<DPP_ROW_NEWBCAST0 + 0>(0.f, fmaf(ra, gin, rg));
            const float afirst = rdpp<DPP_ROW_NEWBCAST0 + 0>(0.f, a[0]);
            if (j == n) { gcar = gout; anx = afirst; }
            // forward pass B: x_i (xs holds b_i on entry)
            float xrun = xseed;
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
                grun = fmaf(alpha, grun, Cn[i]);
                const float gx = grun;
                const float gd = gx * dl[i];
                const float gax = gx * (a[i] * (i == 0 ? xseed : xs[i - 1]));  // g * a_i x_{i-1}
                duv[i] = fmaf(gd, Bn[i], duv[i]);
                ddl[i] = fmaf(gx * Bn[i], uv[i], ddl[i]);
                ddl[i] = fmaf(Araw, gax, ddl[i]);
                dA_loc = fmaf(dl[i], gax, dA_loc);
                accB = __builtin_amdgcn_mfma_f32_16x16x4f32(sel[i], gd * uv[i], accB, 0, 0, 0);
                accC = __builtin_amdgcn_mfma_f32_16x16x4f32(sel[i], dy[i] * xs[i], accC, 0, 0, 0);
            }
            const float dA_tot = row_allsum(dA_loc);
            if (j == n) dAacc += dA_tot;
            // 4-row sums of this state -> LDS (only D rows 0..K-1, i.e. lanes < 4K, carry data)
            if (lane < SL) {
                slab[(((quad * NSPLIT + half) * kMSG + (ni % kMSG)) * 2 + 0) * SL + lane] = accB;
                slab[(((quad * NSPLIT + half) * kMSG + (ni % kMSG)) * 2 + 1) * SL + lane] = accC;
            }
            if (ni % kMSG == kMSG - 1) {
                if (NSPLIT == 2 && ni == kMNS - 1 && half == 1) {  // hand du / ddelta partial sums to the lower half
                    xchg[(quad * 4 + 0) * 64 + lane] = f32x4{duv[0], duv[1], duv[2], duv[3]};
                    xchg[(quad * 4 + 1) * 64 + lane] = f32x4{duv[4], duv[5], duv[6], duv[7]};
                    xchg[(quad * 4 + 2) * 64 + lane] = f32x4{ddl[0], ddl[1], ddl[2], ddl[3]};
                    xchg[(quad * 4 + 3) * 64 + lane] = f32x4{ddl[4], ddl[5], ddl[6], ddl[7]};
                }
                lds_barrier();
                // (half, state, tensor, slab lane) -> sum over the 8 row quads, 4 atomics per thread
                const int t = threadIdx.x;
                if (t < NSPLIT * kMSG * 2 * SL) {
                    const int pl = t % SL, ten = (t / SL) & 1, st = (t / (2 * SL)) % kMSG, hh = t / (2 * SL * kMSG);
                    f32x4 s = slab[(((0 * NSPLIT + hh) * kMSG + st) * 2 + ten) * SL + pl];
#pragma unroll
                    for (int qd = 1; qd < kMQ; ++qd) s += slab[(((qd * NSPLIT + hh) * kMSG + st) * 2 + ten) * SL + pl];
                    const int nn = hh * kMNS + (ni - (kMSG - 1)) + st;
                    const int lo = c * CH + (pl & 15) * K + 4 * (pl >> 4);  // first of 4 positions
                    float* dst = (ten == 0 ? dBg + (int64_t)nn * q.dB_dstate_stride
                                           : dCg + (int64_t)nn * q.dC_dstate_stride) + lo;
                    if (lo < L) {  // seqlen % 8 == 0: the 4 positions are all in or all out
#pragma unroll
                        for (int e = 0; e < 4; ++e) atomicAdd(dst + e, s[e]);
                    }
                }
                if (NSPLIT == 2 && ni == kMNS - 1 && half == 0) {
                    const f32x4 v0 = xchg[(quad * 4 + 0) * 64 + lane], v1 = xchg[(quad * 4 + 1) * 64 + lane];
                    const f32x4 v2 = xchg[(quad * 4 + 2) * 64 + lane], v3 = xchg[(quad * 4 + 3) * 64 + lane];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        duv[i] += v0[i]; duv[4 + i] += v1[i];
                        ddl[i] += v2[i]; ddl[4 + i] += v3[i];
                    }
                }
                lds_barrier();
            }
        };
        // rolled (by 2) on purpose: a fully unrolled state loop does not fit the instruction cache
        RawVec<T, K> rawB1, rawC1;
#pragma unroll 1
        for (int ni = 0; ni < kMNS; ni += 2) {
            do_state(ni, rawB, rawC, rawB1, rawC1);
            do_state(ni + 1, rawB1, rawC1, rawB, rawC);
        }
        if (half == 0) {
            float raw[K];
            RawVec<T, K> t0;
            t0.load(dt_b, VMS_OFF(p.delta_batch_stride, p.delta_d_stride) + l0, ok);
            t0.widen(raw);
#pragma unroll
            for (int i = 0; i < K; ++i) {
                if (p.delta_softplus) {
                    const float rr = raw[i] + bias;
                    ddl[i] = rr <= 20.f ? ddl[i] * sigmoidf_(rr) : ddl[i];
                }
                dbias_acc += ok ? ddl[i] : 0.f;
            }
            if (ok) {
                store_blocked<T, K, true>(du_b + (VMS_OFF(q.du_batch_stride, q.du_d_stride) + l0), K, duv);
                store_blocked<T, K, true>(ddelta_b + (VMS_OFF(q.ddelta_batch_stride, q.ddelta_d_stride) + l0), K, ddl);
            }
        }
    }
#undef VMS_OFF
    const float dD_tot = row_allsum(dD_acc), db_tot = row_allsum(dbias_acc);
    if (row_ok) {
        if (half == 0) {
            if (q.dD && j == 0) atomicAdd(q.dD + d, dD_tot);
            if (q.ddelta_bias && j == 0) atomicAdd(q.ddelta_bias + d, db_tot);
        }
        if (j / kMNS == half) atomicAdd(q.dA + (int64_t)d * q.dA_d_stride + (int64_t)j * q.dA_dstate_stride, dAacc);
    }
}

bool scan_bwd_mfma_eligible(const vms_scan_bwd_params& q, bool vec) {
    const vms_scan_fwd_params& p = q.f;
    if (!vec || !p.is_variable_B || !p.is_variable_C || p.dstate != kMN || !p.x || p.x_has_sub != 1) return false;
    const int dpg = p.dim / p.n_groups;
    if (dpg % kMRows != 0) return false;     // a workgroup's rows must share one B/C group
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

template <typename T>
static int launch_mfma(const vms_scan_bwd_params& q, hipStream_t stream) {
    constexpr int NSPLIT = VMS_BWD_NSPLIT;
    const vms_scan_fwd_params& p = q.f;
    const int tiles = (p.dim + kMRows - 1) / kMRows;
    dim3 grid(p.batch * tiles), block(kMQ * NSPLIT * kWave);
    // slabs (+ du/ddelta exchange between the state halves)
    const size_t smem = 16 * (kMQ * NSPLIT * kMSG * 2 * (4 * kMK) + (NSPLIT == 2 ? kMQ * 4 * 64 : 0));
    // 96 KB of dynamic LDS: above the 64 KB default limit, must be allowed per kernel and per device
    static PerDeviceOnce attr_once;
    const hipError_t arc = attr_once.run([&]() -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&scan_bwd_mfma_kernel<T, true, NSPLIT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&scan_bwd_mfma_kernel<T, false, NSPLIT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        return e;
    });
    if (arc != hipSuccess) {
        set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed: %s", hipGetErrorString(arc));
        return VMS_ERR_LAUNCH;
    }
    if (p.z) hipLaunchKernelGGL((scan_bwd_mfma_kernel<T, true, NSPLIT>), grid, block, smem, stream, q);
    else hipLaunchKernelGGL((scan_bwd_mfma_kernel<T, false, NSPLIT>), grid, block, smem, stream, q);
    VMS_LAUNCH_CHECK();
    return VMS_OK;
}

int launch_scan_bwd_mfma(const vms_scan_bwd_params& q, hipStream_t stream) {
    switch (q.f.dtype) {
        case VMS_BF16: return launch_mfma<bf16_t>(q, stream);
        case VMS_F16: return launch_mfma<f16_t>(q, stream);
        default: return launch_mfma<float>(q, stream);
    }
}

}  // namespace vms
