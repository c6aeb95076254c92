// selective_scan_bwd_pair.hip -- backward selective scan for gfx950 (wave64), ELEMENT PAIRS in packed fp32.
//
// Replaces selective_scan_bwd_kernel (mamba/csrc/selective_scan/selective_scan_bwd_kernel.cuh:75-489) for
// variable B/C, dstate 16, 16-byte aligned I/O, seqlen % 8 == 0, with the forward's 128-element
// sub-checkpoints in x (vms_hip.h, x_has_sub == 1).  Same decomposition as selective_scan_bwd_mfma.hip:
//   * a wave owns 4 rows: lane = 16*r + j, row r = one DPP row, lane j owns the 8 consecutive elements
//     [8j, 8j+8) of the row's current 128-element chunk; chunks are walked from the end of the sequence;
//     the forward re-scan and the adjoint suffix scan are 4-step DPP row scans;
//   * B / C of a chunk are shared by the 32 rows of a workgroup: they are fetched once per workgroup (one
//     16-byte vector per thread), widened to fp32 and parked in LDS (double-buffered across chunks); every
//     lane then reads its 8 + 8 values per state with four ds_read_b128 -- no per-wave widening;
//   * dB / dC: every lane writes its 8 + 8 products to an LDS slab [state % 2][tensor][row][position];
//     every second state the workgroup sums the 32 rows (16 ds_read_b64 + 15 v_pk_add per thread, the two
//     row halves joined by one v_permlane32_swap) and issues ONE fp32 atomic per (state, tensor, position);
//     this takes the row reduction off the VALU critical path (in-wave permlane trees cost 24 ops per state);
//   * per-(row, state) carries live in one register (lane j <-> state j), handed out with ds_bpermute.
// What is new here: everything that is independent across the 8 elements of a lane (exp arguments, b = delta u B,
// c = C dy, g a x, the du / ddelta / dA / dB / dC contributions) is computed on PAIRS of consecutive elements
// with v_pk_mul_f32 / v_pk_fma_f32; only the four recurrences along the elements stay scalar chains.  This
// kernel has 2048 waves at (8, 8192, 1024, 16) = 2 per SIMD, and a wave issues one instruction every ~8.5
// cycles whatever it is (tools/microbench/microbench.hip): instruction COUNT is what the run time follows, and the
// pairing takes it from ~32 to ~20 per (element, state) without growing the register footprint (pairing
// the STATES instead needs 64-bit versions of every per-element array: 350 VGPRs, measured slower).
// du / ddelta are accumulated as S1_i = sum_n g B, S2_i = sum_n A g a x_{i-1} and combined once per chunk;
// the lane aggregates' "a" components come from exp2(A * sum(delta)) instead of running products.
// `reverse` (vms_hip.h) is supported: the lane's 8 logical elements are read / written right-to-left.
#include "vms_common.h"
#include "scan_bwd_helpers.h"

namespace vms {

// RAG: seqlen % 8 != 0 -- the last valid lane of a row owns nv < 8 elements: its activations move one by one
// (once per row), masks are per element, B / C come through the caller's padding (vms_hip.h bc_pad).
template <typename T, bool HZ, bool REV, bool RAG>
__global__ __launch_bounds__(kBQ* kWave) void scan_bwd_pair_kernel(const vms_scan_bwd_params q, const int n_seg,
                                                                       const float2* __restrict__ seg_carry) {
    const vms_scan_fwd_params& p = q.f;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int K = kBK, N = kBN;
    constexpr int CH = kCH;      // elements per row per iteration (128)
    // LDS: fp32 B / C of two chunks, then the dB / dC slab.  A row's 128 positions are stored as
    // [half][lane j][4]: element i of lane j sits at float (i / 4) * 64 + 4 j + i % 4, so that the 16-byte
    // accesses of consecutive lanes are consecutive (no bank conflicts).
    lds_f4* const bc4 = (lds_f4*)smem;
    lds_f4* const slab4 = (lds_f4*)(smem + kBcFloats);
    lds_f2* const slab2 = (lds_f2*)(smem + kBcFloats);
    const int lane = threadIdx.x & 63;
    const int quad = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, r = lane >> 4;
    // per (row, state) record {A, state entering the chunk, a of the first element to the right, adjoint
    // entering from the right}: wave-private, read one state ahead with a single ds_read_b128 (all 16 lanes of
    // a row read the same address), the two carries rewritten by the row's lane 0 at the end of each state
    lds_f4* const rec4 = (lds_f4*)(smem + kBcFloats + 2 * kSlabFloats) + (quad * 4 + (lane >> 4)) * kBN;
    __attribute__((address_space(3))) float* const rec1 = (__attribute__((address_space(3))) float*)rec4;
    // consecutive workgroups share a row tile across batches -> batch = blockIdx % batch keeps the
    // B/C of one batch on one XCD's L2 when batch == 8
    // n_seg > 1 (few rows, long sequences): the grid repeats n_seg times and copy `seg` walks only the chunks
    // [c_lo, c_hi) of every row; the adjoint entering that range from the right comes from scan_bwd_carry_kernel
    const int wg_per_seg = gridDim.x / n_seg;
    const int seg = blockIdx.x / wg_per_seg, wg = blockIdx.x - seg * wg_per_seg;
    const int b = wg % p.batch;
    const int d0 = (wg / p.batch) * kBRows;
    const int d = d0 + quad * 4 + r;
    const bool row_ok = d < p.dim;
    const int dc = row_ok ? d : p.dim - 1;
    const int g = d0 / (p.dim / p.n_groups);  // host guarantees one group per workgroup
    const int L = p.seqlen;

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
    const T* const Bv = static_cast<const T*>(p.B) + (int64_t)b * p.B_batch_stride + (int64_t)g * p.B_group_stride;
    const T* const Cv = static_cast<const T*>(p.C) + (int64_t)b * p.C_batch_stride + (int64_t)g * p.C_group_stride;
    float* const dBg = q.dB + (int64_t)b * q.dB_batch_stride + (int64_t)g * q.dB_group_stride;
    float* const dCg = q.dC + (int64_t)b * q.dC_batch_stride + (int64_t)g * q.dC_group_stride;
    const float* const x_b = static_cast<const float*>(p.x);
    const float Dd = p.D ? static_cast<const float*>(p.D)[dc] : 0.f;
    const float bias = p.delta_bias ? static_cast<const float*>(p.delta_bias)[dc] : 0.f;
    // A of the lane's row: lane j keeps A[d][j]; handed out per state by a row broadcast
    const float A_mine = static_cast<const float*>(p.A)[(int64_t)dc * p.A_d_stride + (int64_t)j * p.A_dstate_stride];

    float dAacc = 0.f;  // dA[d][j]
    float dD_acc = 0.f, dbias_acc = 0.f;

    // B / C staging: thread (tensor, state, j) = (wave >> 2, 4 (wave & 3) + DPP row, lane & 15) fetches the 8
    // values [8j, 8j+8) of one state of the NEXT chunk while the current one computes
    RawB<T, REV> stg;
    const int st_ten = quad >> 2, st_n = (quad & 3) * 4 + r;
    const T* const st_src = (st_ten ? Cv + (int64_t)st_n * p.C_dstate_stride : Bv + (int64_t)st_n * p.B_dstate_stride);
    bool st_ok = false;
    auto stage_issue = [&](int cc) __attribute__((always_inline)) {
        const int ll = cc * CH + j * K;
        st_ok = cc >= 0 && ll < L;
        if (RAG) stg.load_s(st_src, REV ? L - ll - K : ll, st_ok, REV ? L - K : 0);
        else stg.load(st_src, REV ? L - ll - K : ll, st_ok);
    };
    auto stage_commit = [&]() __attribute__((always_inline)) {
        f32x4 lo, hi;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            lo[i] = st_ok ? stg.at(i) : 0.f;
            hi[i] = st_ok ? stg.at(4 + i) : 0.f;
        }
        lds_f4* dst = bc4 + ((st_ten * N + st_n) * CH) / 4 + j;
        dst[0] = lo;
        dst[16] = hi;
    };
    // per-state fp32 B / C of the lane: two register sets (current state, next state)
    struct BCReg { f32x4 b0, b1, c0, c1; };
    auto bc_fetch = [&](BCReg& t, int n) __attribute__((always_inline)) {
        const lds_f4* src = bc4 + (n * CH) / 4 + j;
        t.b0 = src[0];
        t.b1 = src[16];
        t.c0 = src[N * CH / 4];
        t.c1 = src[N * CH / 4 + 16];
    };
    BCReg bcA, bcB;
    // row reduction of the slab: wave -> (state % 2, tensor, half of the position pairs), lanes 0..31 / 32..63
    // sum rows 0..15 / 16..31 of one position pair; after the swap lane l < 32 owns float 2m, lane l + 32
    // float 2m + 1 of the swizzled row
    const int rd_st = quad >> 2, rd_ten = (quad >> 1) & 1;
    const int rd_m = (quad & 1) * 32 + (lane & 31);
    const lds_f2* const rd_src = slab2 + (((rd_st * 2 + rd_ten) * kBRows + (lane >> 5) * 16) * CH) / 2 + rd_m;
    // the sums of a pair are taken while the NEXT pair computes (slab double-buffered): half of the 16 reads
    // are issued at the start of each of its two states and consumed at the end of that state
    f2 rdv[8], racc = f2{0.f, 0.f};
    float* rd_ptr = nullptr;  // where the pair being summed goes
    bool rd_okp = false;
    const int rd_idx = 2 * rd_m + (lane >> 5);
    const int rd_pos = 8 * ((rd_idx & 63) >> 2) + 4 * (rd_idx >> 6) + (rd_idx & 3);   // position inside the chunk
    float* const rd_dst = rd_ten ? dCg : dBg;
    const int64_t rd_stride = rd_ten ? q.dC_dstate_stride : q.dB_dstate_stride;
    // the row's own data of the NEXT chunk is requested while the current one computes
    RawB<T, REV> pu, pdt, pdo, pz, pout;
    float hck_next = 0.f;
    const int n_c = (L + CH - 1) / CH;
    const uint32_t o_x = p.x ? static_cast<uint32_t>(((int64_t)b * p.dim + dc) * p.n_chunks * p.x_chunk_stride) : 0u;
    auto request_row = [&](int cc) __attribute__((always_inline)) {
        const int ll = cc * CH + j * K;
        const bool v = cc >= 0 && ll < L && row_ok;
        const uint32_t pl = REV ? L - ll - K : ll;
        if (!RAG || !v || L - ll >= K) {
            pu.load(u_b, VMS_OFF(p.u_batch_stride, p.u_d_stride) + pl, v);
            pdt.load(dt_b, VMS_OFF(p.delta_batch_stride, p.delta_d_stride) + pl, v);
            pdo.load(dout_b, VMS_OFF(q.dout_batch_stride, q.dout_d_stride) + pl, v);
            if (HZ) {
                pz.load(z_b, VMS_OFF(p.z_batch_stride, p.z_d_stride) + pl, v);
                pout.load(outp_b, VMS_OFF(p.out_batch_stride, p.out_d_stride) + pl, v);
            }
        } else {  // the row's last, partly valid vector
            const int nvn = L - ll;
            pu.load_partial(u_b + VMS_OFF(p.u_batch_stride, p.u_d_stride), ll, L, nvn);
            pdt.load_partial(dt_b + VMS_OFF(p.delta_batch_stride, p.delta_d_stride), ll, L, nvn);
            pdo.load_partial(dout_b + VMS_OFF(q.dout_batch_stride, q.dout_d_stride), ll, L, nvn);
            if (HZ) {
                pz.load_partial(z_b + VMS_OFF(p.z_batch_stride, p.z_d_stride), ll, L, nvn);
                pout.load_partial(outp_b + VMS_OFF(p.out_batch_stride, p.out_d_stride), ll, L, nvn);
            }
        }
        // state entering chunk cc = 128-element sub-checkpoint cc-1 (vms_hip.h); lane j loads state j
        const int e128 = cc * (CH / 128) - 1;
        const uint32_t xo = cc > 0 ? o_x + x_sub_off(e128, j, (int)p.x_chunk_stride, p.x_has_sub == 3, kBN) : 0u;
        hck_next = x_b[xo];
    };
    const int cps = (n_c + n_seg - 1) / n_seg;                        // chunks per segment
    const int c_lo = seg * cps, c_hi = (c_lo + cps < n_c) ? c_lo + cps : n_c;
    float g_in = 0.f, anx_in = 1.f;                                   // lane j <-> state j
    if (seg < n_seg - 1) {
        const float2* cp = seg_carry + (((int64_t)b * p.dim + dc) * n_seg) * N + j;
        for (int s2 = n_seg - 1; s2 > seg; --s2) {                    // g entering segment s = P_{s+1} g_{s+1} + q_{s+1}
            const float2 pq = cp[(int64_t)s2 * N];
            g_in = fmaf(pq.x, g_in, pq.y);
        }
        const int lr = c_hi * CH;                                     // first element to the right of this range
        float t = static_cast<float>(dt_b[VMS_OFF(p.delta_batch_stride, p.delta_d_stride) + (REV ? L - 1 - lr : lr)]) + bias;
        if (p.delta_softplus) t = softplusf_(t);
        anx_in = fast_exp2(t * A_mine * kLog2e);
    }
    request_row(c_hi - 1);
    stage_issue(c_hi - 1);
    stage_commit();
    rec4[j] = f32x4{A_mine, c_hi > 1 ? hck_next : 0.f, anx_in, g_in};   // lane j <-> state j
    lds_barrier_b();
    bc_fetch(bcA, 0);
    f32x4 bc = rec4[0];   // record of the state about to run
    const bool is_first = j == 0, is_last = j == 15;
    for (int c = c_hi - 1; c >= c_lo; --c) {
        const int l0 = c * CH + j * K;
        const bool okb = l0 < L, ok = okb && row_ok;
        const int nv = !ok ? 0 : (RAG && L - l0 < K ? L - l0 : K);   // valid elements of the lane
        const bool full = !RAG || nv == K || nv == 0;
        const uint32_t pl0 = REV ? L - l0 - K : l0;      // physical start of the lane's K elements
        const int rd_lo = c * CH + rd_pos;
        float* const rd_dst_c = rd_dst + (REV ? L - 1 - rd_lo : rd_lo);
        float uv[K], dy[K];
        f2 dl2[K / 2], dlu2[K / 2], dy2[K / 2], sg2[K / 2];  // sg = d softplus / d(delta + bias)
        float sdl = 0.f, dl_first = 0.f;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            uv[i] = pu.at(i);
            dy[i] = (RAG ? i < nv : ok) ? pdo.at(i) : 0.f;  // past the end: c = 0, a = 1 (identity for the suffix scan)
            float t = pdt.at(i) + bias, sg = 1.f;
            if (p.delta_softplus) {  // selective_scan_fwd_kernel.cuh:153-156 and bwd_kernel.cuh:439-452
                const float e = fast_exp(t);
                const float w = 1.f + e;
                const float rw = fast_rcp(w);
                const float sp = fmaf(e - (w - 1.f), rw, fast_log(w));  // log1p(e), see softplusf_
                sg = t <= 20.f ? e * rw : 1.f;
                t = t <= 20.f ? sp : t;
            }
            t = (RAG ? i < nv : ok) ? t : 0.f;
            dl2[i / 2][i % 2] = t;
            sg2[i / 2][i % 2] = sg;
            sdl += t;
            if (i == 0) dl_first = t;
        }
        if (HZ) {
            float zv[K], ov[K], dzv[K];
#pragma unroll
            for (int i = 0; i < K; ++i) { zv[i] = pz.at(i); ov[i] = pout.at(i); }
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const float s = sigmoidf_(zv[i]);
                const float silu = zv[i] * s;
                dzv[i] = dy[i] * ov[i] * s * (1.f + zv[i] * (1.f - s));
                dy[i] *= silu;
                ov[i] *= silu;
            }
            if (!full) {  // the row's last, partly valid vector
                T* const dzr = dz_b + VMS_OFF(q.dz_batch_stride, q.dz_d_stride);
                if (q.dz_accumulate) {
                    RawB<T, REV> od;
                    od.load_partial(dzr, l0, L, nv);
#pragma unroll
                    for (int i = 0; i < K; ++i) dzv[i] += od.at(i);
                }
                store_partial_b<T, REV>(dzr, l0, L, nv, dzv);
                if (out_z_b) store_partial_b<T, REV>(out_z_b + VMS_OFF(p.out_z_batch_stride, p.out_z_d_stride), l0, L, nv, ov);
            } else if (ok) {
                if (q.dz_accumulate) {  // dz += (vms_hip.h)
                    RawB<T, REV> od;
                    od.load(dz_b, VMS_OFF(q.dz_batch_stride, q.dz_d_stride) + pl0, true);
#pragma unroll
                    for (int i = 0; i < K; ++i) dzv[i] += od.at(i);
                }
                store_b<T, REV>(dz_b + (VMS_OFF(q.dz_batch_stride, q.dz_d_stride) + pl0), dzv);
                if (out_z_b) store_b<T, REV>(out_z_b + (VMS_OFF(p.out_z_batch_stride, p.out_z_d_stride) + pl0), ov);
            }
        }
        f2 S1[K / 2], S2[K / 2];  // per element: sum_n g B  /  sum_n A g a x_{i-1}
#pragma unroll
        for (int i = 0; i < K; ++i) {
            dy2[i / 2][i % 2] = dy[i];
            dlu2[i / 2][i % 2] = dl2[i / 2][i % 2] * uv[i];
            dD_acc = fmaf(dy[i], uv[i], dD_acc);
        }
#pragma unroll
        for (int k = 0; k < K / 2; ++k) {
            S1[k] = f2{0.f, 0.f};
            S2[k] = f2{0.f, 0.f};
        }
        request_row(c - 1);  // in flight during the 16 states of this chunk
        stage_issue(c - 1);
#define VMS_EL(arr, i) arr[(i) / 2][(i) % 2]
        // One state.  Everything that is independent across the lane's 8 elements runs on ELEMENT PAIRS
        // (v_pk_*): widening products, the gradient contributions; the four recurrences along the elements
        // (forward / adjoint, lane aggregate / seeded) stay scalar chains.
        auto do_state = [&](const int n, const int i4, const BCReg& cur, BCReg& nxt) __attribute__((always_inline)) {
            const int par = i4 & 1, buf = (i4 >> 1) & 1;   // state of the pair, slab buffer of the pair
            if (n + 1 < N) bc_fetch(nxt, n + 1);          // B / C of the next state
            {   // rows [8 par, 8 par + 8) of this thread's half of the previous pair
                const lds_f2* src = rd_src + ((buf ^ 1) * kSlabFloats) / 2 + par * 8 * (CH / 2);
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) rdv[rr] = src[rr * (CH / 2)];
            }
            const float Araw = bc.x, hin = bc.y, anx_n = bc.z, gin = bc.w;
            // the next state's record (waited for at the end of this state, see the anchor below); before the
            // last state's read, lane j stores the state entering the NEXT chunk for state j
            if (i4 == 3 && n == N - 1) rec1[4 * j + 1] = c > 1 ? hck_next : 0.f;
            bc = rec4[(n + 1) & (N - 1)];
            const float An = Araw * kLog2e;
            const f2 An2 = f2{An, An}, Araw2 = f2{Araw, Araw};
            f2 Bn2[K / 2], c2[K / 2], a2[K / 2], xs2[K / 2], ax2[K / 2];
            // ---- local scans: forward (a, b) and adjoint (alpha = a_{i+1}, c = C dy) ----
#pragma unroll
            for (int k = 0; k < K / 2; ++k) {
                Bn2[k] = k == 0 ? f2{cur.b0.x, cur.b0.y} : k == 1 ? f2{cur.b0.z, cur.b0.w} : k == 2 ? f2{cur.b1.x, cur.b1.y} : f2{cur.b1.z, cur.b1.w};
                const f2 t = dl2[k] * An2;
                a2[k] = f2{fast_exp2(t.x), fast_exp2(t.y)};
                xs2[k] = dlu2[k] * Bn2[k];  // b_i for now
                c2[k] = (k == 0 ? f2{cur.c0.x, cur.c0.y} : k == 1 ? f2{cur.c0.z, cur.c0.w} : k == 2 ? f2{cur.c1.x, cur.c1.y} : f2{cur.c1.z, cur.c1.w}) * dy2[k];
            }
            float px = 0.f;
#pragma unroll
            for (int i = 0; i < K; ++i) px = fmaf(VMS_EL(a2, i), px, VMS_EL(xs2, i));
            float pa = fast_exp2(sdl * An);
            const float a_right = bdpp<DPP_ROW_SHL1>(anx_n, a2[0].x);  // lane 15 of the row <- next chunk
            float rg = 0.f;
#pragma unroll
            for (int i = K - 1; i >= 0; --i) rg = fmaf(i == K - 1 ? a_right : VMS_EL(a2, i + 1), rg, VMS_EL(c2, i));
            float ra = fast_exp2((sdl - dl_first) * An) * a_right;
            // the carries enter at the end lanes, so that the inclusive scans deliver the true states / adjoints
            px = fmaf(pa, is_first ? hin : 0.f, px);
            rg = fmaf(ra, is_last ? gin : 0.f, rg);
            row_scan_pair_b(pa, px, ra, rg);
            const float xseed = bdpp<DPP_ROW_SHR1>(hin, px);  // state entering this lane's first element
            float grun = bdpp<DPP_ROW_SHL1>(gin, rg);         // adjoint entering this lane's last element
            // new carries = lane 0's a of its first element and the adjoint leaving it
            if (is_first) *(lds_f2*)(rec1 + 4 * n + 2) = f2{a2[0].x, rg};
            // forward pass B: a_i x_{i-1} and x_i (xs2 holds b_i on entry)
            {
                float xrun = xseed;
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    const float ax = VMS_EL(a2, i) * xrun;
                    xrun = ax + VMS_EL(xs2, i);
                    VMS_EL(ax2, i) = ax;
                    VMS_EL(xs2, i) = xrun;
                }
            }
            // adjoint pass B: g_i (c2 is overwritten)
#pragma unroll
            for (int i = K - 1; i >= 0; --i) {
                grun = fmaf(i == K - 1 ? a_right : VMS_EL(a2, i + 1), grun, VMS_EL(c2, i));
                VMS_EL(c2, i) = grun;
            }
            // per-element contributions, two elements per instruction
            f2 dA2 = f2{0.f, 0.f};
            f2 dBv[K / 2], dCv[K / 2];
#pragma unroll
            for (int k = 0; k < K / 2; ++k) {
                const f2 g2 = c2[k];
                const f2 gax = g2 * ax2[k];  // g * a_i x_{i-1}
                S1[k] = pk_fma_b(g2, Bn2[k], S1[k]);
                S2[k] = pk_fma_b(Araw2, gax, S2[k]);
                dA2 = pk_fma_b(dl2[k], gax, dA2);
                dBv[k] = g2 * dlu2[k];
                dCv[k] = dy2[k] * xs2[k];
            }
            const float dA_tot = row_allsum_b(dA2.x + dA2.y);
            if (j == n) dAacc += dA_tot;
            // everything requested at the start of the state has arrived by now
            asm volatile(""
                         : "+v"(bc), "+v"(rdv[0]), "+v"(rdv[1]), "+v"(rdv[2]),
                           "+v"(rdv[3]), "+v"(rdv[4]), "+v"(rdv[5]), "+v"(rdv[6]), "+v"(rdv[7]));
            {
                const f2 t = ((rdv[0] + rdv[1]) + (rdv[2] + rdv[3])) + ((rdv[4] + rdv[5]) + (rdv[6] + rdv[7]));
                racc = par == 0 ? t : racc + t;
            }
            // this row's products -> slab[buf][par][tensor][row]
            {
                lds_f4* dst = slab4 + (buf * kSlabFloats + ((par * 2 + 0) * kBRows + quad * 4 + r) * CH) / 4 + j;
                dst[0] = __builtin_shufflevector(dBv[0], dBv[1], 0, 1, 2, 3);
                dst[16] = __builtin_shufflevector(dBv[2], dBv[3], 0, 1, 2, 3);
                dst[kBRows * CH / 4] = __builtin_shufflevector(dCv[0], dCv[1], 0, 1, 2, 3);
                dst[kBRows * CH / 4 + 16] = __builtin_shufflevector(dCv[2], dCv[3], 0, 1, 2, 3);
            }
            if (par == 1) {
                float sx = racc.x, sy = racc.y;
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(sx), "+v"(sy));
                if (rd_okp) atomicAdd(rd_ptr, sx + sy);
                // the pair just written is summed during the next one
                rd_ptr = rd_dst_c + (int64_t)(n - 1 + rd_st) * rd_stride;
                rd_okp = rd_lo < L;
                lds_barrier_b();  // pair written by all rows; previous pair's buffer free again
            }
        };
        // rolled (by 4) on purpose: a fully unrolled state loop does not fit the instruction cache
#pragma unroll 1
        for (int n = 0; n < N; n += 4) {
            do_state(n, 0, bcA, bcB);
            do_state(n + 1, 1, bcB, bcA);
            do_state(n + 2, 2, bcA, bcB);
            do_state(n + 3, 3, bcB, bcA);
        }
        // every wave is past its last B / C read of this chunk (barrier of the last pair): swap in the next chunk's
        stage_commit();
        lds_barrier_b();
        bc_fetch(bcA, 0);
#undef VMS_EL
        {
            float duv[K], ddl[K];
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const float s1 = S1[i / 2][i % 2], s2 = S2[i / 2][i % 2];
                duv[i] = fmaf(dl2[i / 2][i % 2], s1, Dd * dy2[i / 2][i % 2]);
                ddl[i] = fmaf(uv[i], s1, s2);
                ddl[i] *= sg2[i / 2][i % 2];
                dbias_acc += (RAG ? i < nv : ok) ? ddl[i] : 0.f;
            }
            if (!full) {
                store_partial_b<T, REV>(du_b + VMS_OFF(q.du_batch_stride, q.du_d_stride), l0, L, nv, duv);
                store_partial_b<T, REV>(ddelta_b + VMS_OFF(q.ddelta_batch_stride, q.ddelta_d_stride), l0, L, nv, ddl);
            } else if (ok) {
                store_b<T, REV>(du_b + (VMS_OFF(q.du_batch_stride, q.du_d_stride) + pl0), duv);
                store_b<T, REV>(ddelta_b + (VMS_OFF(q.ddelta_batch_stride, q.ddelta_d_stride) + pl0), ddl);
            }
        }
    }
#undef VMS_OFF
    {   // the last pair (buffer 1) is still in the slab
        const lds_f2* src = rd_src + kSlabFloats / 2;
        f2 t = src[0];
#pragma unroll
        for (int rr = 1; rr < 16; ++rr) t += src[rr * (CH / 2)];
        float sx = t.x, sy = t.y;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(sx), "+v"(sy));
        if (rd_okp) atomicAdd(rd_ptr, sx + sy);
    }
    const float dD_tot = row_allsum_b(dD_acc), db_tot = row_allsum_b(dbias_acc);
    if (row_ok) {
        if (q.dD && j == 0) atomicAdd(q.dD + d, dD_tot);
        if (q.ddelta_bias && j == 0) atomicAdd(q.ddelta_bias + d, db_tot);
        atomicAdd(q.dA + (int64_t)d * q.dA_d_stride + (int64_t)j * q.dA_dstate_stride, dAacc);
    }
}


// =====================================================================================================================
// Second generation for whole-vector rows (seqlen % 8 == 0; ragged rows stay on scan_bwd_pair_kernel above): the same
// decomposition and arithmetic, reorganised around what the SQ counters of that kernel show
// (profiles/r02a_sq_scan.md: 6.4 cycles per VALU issue unit against 4.5 for the forward kernel, a third of the wave
// cycles parked in s_waitcnt / s_barrier).  The grid of the benchmark shape is 2,048 waves = 2 per SIMD whatever the
// register count, so the lever is not occupancy but what the two waves of a SIMD wait for:
//   * workgroups of W = 4 waves (16 rows) with 36 KB of LDS: the two waves of a SIMD belong to DIFFERENT workgroups,
//     so a workgroup at its barrier (one per pair of states) or waiting for its chunk's loads leaves the SIMDs to
//     its neighbour, and a barrier synchronises 4 waves instead of 8;
//   * dB / dC: the 4 rows of a wave are summed IN the wave first -- v_permlane32_swap pairs dB[e] with dC[e]
//     (lanes 0-31 then hold dB, lanes 32-63 dC, each summed over rows r and r+2), v_permlane16_swap pairs elements
//     e and e+4 (DPP row rho then holds tensor rho >> 1, elements 4 (rho & 1) .. +3, summed over all 4 rows): 12
//     swaps + 6 v_pk_add_f32, and ONE ds_write_b128 per lane and state.  The slab holds one wave partial per position
//     instead of 4 rows (1 KB per wave and state), every thread sums the W partials of its (state, tensor,
//     position)s with 8 ds_read_b32 per pair and issues the atomics;
//   * no second B / C register set (the fp32 B / C of a state are read from LDS when the state starts), the
//     softplus derivative sigmoid(delta_raw) is carried from the prologue (8 registers; round 3), u is
//     widened again from its raw vector in the epilogue; the next chunk's row data and B / C pieces are requested
//     after the last state, into registers the state temporaries just vacated (182 VGPRs instead of 250).
// Measured in round 5 and removed from the source (profiles/r05_scan_ablations.md; the code: profiles/r05_bwd_rowsum_variants.patch): the 4-row
// dB / dC sums through wave-private LDS (VALU time -14 %, run time +5-6 % even with the reads a state behind the writes) and on the matrix pipe
// (two v_mfma_f32_16x16x32_bf16 with one-hot row selectors: run time -6 %, but the bf16-rounded products leave the reference's element-wise
// tolerance at small row counts).
// timing-only ablations (wrong results): tools/variant.sh <tag> -DVMS_ABL_NOBAR=1 / -DVMS_ABL_NOATOM=1
#ifndef VMS_ABL_NOBAR
#define VMS_ABL_NOBAR 0
#endif
#ifndef VMS_ABL_NOATOM
#define VMS_ABL_NOATOM 0
#endif
#ifndef VMS_BWD_PRIO
#define VMS_BWD_PRIO 1
#endif
// states between two workgroup barriers = states per slab buffer.  8-wave workgroups: 4 (round 5; 2 before: 16 -> 8 barriers per chunk,
// the dual call at (8, 1024, 8192) 1,533 -> 1,510 us, 64 KB of slab; 8 needs 128 KB and spills: 2,100 us); 4-wave workgroups: 2 (4 measured
// slower at (8, 768, 3136), their barriers already leave the SIMD to the other workgroup).  profiles/r05_scan_ablations.md
#ifndef VMS_BWD_SG
#define VMS_BWD_SG 4
#endif
#ifndef VMS_BWD_SG4
#define VMS_BWD_SG4 2
#endif
template <int W> struct B4 {
    static constexpr int kSG = W == 8 ? VMS_BWD_SG : VMS_BWD_SG4;   // states between two workgroup barriers (= states per slab buffer)
    static constexpr int kPair = kSG * W * 4 * kWave;   // floats of one group of states: [state][wave][4 lane + k]
    static constexpr int kRows = 4 * W;
    static constexpr int kPPT = 8 / W;                 // B / C pieces and slab outputs per thread (512 per chunk / pair)
    // per-(row, state) records: 16 records of 16 bytes per row + one record of padding -- at a 256-byte pitch the four rows of a
    // wave sit on the same banks, and the record read every state makes (16 lanes of a row, one address) was a 4-way conflict
    static constexpr int kRecPitch = kBN + 1;
    static constexpr size_t kSmem = sizeof(float) * (kBcFloats + 2 * kPair + kRows * kRecPitch * 4);
};

// DZM (two directions of a bidirectional block in one launch, vms_selective_scan_bwd_dual): 0 = dz from this launch's own
// `out` (the single-direction call); 1 = dz = dout (out + out2) dsilu(z) -- the gradient z receives through BOTH directions,
// which is linear in the pre-gate outputs (out2 = the other direction's, same physical positions); 2 = no dz (the other
// direction's workgroups write it).  bid / nblk: this problem's workgroup index and count inside the launch.
// NT: dstate as a compile-time value (16: the tuned instantiations) or 0 = p.dstate in {4, 8} at run time (round 6; W = 4 only: its
// 2-state slab groups give an even number of groups per chunk for every such N, which the double-buffered slab needs).  The LDS
// layouts keep their 16-state pitches; states >= N are neither staged nor walked, and the lanes j >= N of a row -- which hold A, the
// carried adjoint and dA of state j -- touch no memory.
template <typename T, bool HZ, bool REV, int W, bool XL, int DZM = 0, int NT = kBN>
__device__ __forceinline__ void scan_bwd_pair4_body(const vms_scan_bwd_params& q, const int n_seg, const float2* __restrict__ seg_carry,
                                                    const int bid, const int nblk, const T* __restrict__ out2_b = nullptr,
                                                    const int64_t out2_batch_stride = 0, const int64_t out2_d_stride = 0,
                                                    const int sub_m = 1) {
    const vms_scan_fwd_params& p = q.f;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int K = kBK, CH = kCH;
    static_assert(NT == kBN || (NT == 0 && W == 4), "run-time dstate: 4-wave workgroups only");
    const int N = NT ? NT : p.dstate;
    constexpr int kB4Pair = B4<W>::kPair, PPT = B4<W>::kPPT, kRows4 = B4<W>::kRows;
    lds_f4* const bc4 = (lds_f4*)smem;                                   // fp32 B / C of the chunk: [tensor][state][128]
    lds_f4* const slab4 = (lds_f4*)(smem + kBcFloats);                   // [buf][par][wave][lane] float4
    const lds_f32* const slab1 = (const lds_f32*)(smem + kBcFloats);
    const int lane = threadIdx.x & 63;
    const int quad = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, r = lane >> 4;
    lds_f4* const rec4 = (lds_f4*)(smem + kBcFloats + 2 * kB4Pair) + (quad * 4 + r) * B4<W>::kRecPitch;
    __attribute__((address_space(3))) float* const rec1 = (__attribute__((address_space(3))) float*)rec4;
    const int wg_per_seg = nblk / n_seg;
    const int seg = bid / wg_per_seg, wg = bid - seg * wg_per_seg;
    const int b = wg % p.batch;
    const int d0 = (wg / p.batch) * kRows4;
    const int d = d0 + quad * 4 + r;
    const bool row_ok = d < p.dim;
    const int dc = row_ok ? d : p.dim - 1;
    const int g = d0 / (p.dim / p.n_groups);  // host guarantees one group per workgroup
    const int L = p.seqlen;

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
    const T* const Bv = static_cast<const T*>(p.B) + (int64_t)b * p.B_batch_stride + (int64_t)g * p.B_group_stride;
    const T* const Cv = static_cast<const T*>(p.C) + (int64_t)b * p.C_batch_stride + (int64_t)g * p.C_group_stride;
    float* const dBg = q.dB + (int64_t)b * q.dB_batch_stride + (int64_t)g * q.dB_group_stride;
    float* const dCg = q.dC + (int64_t)b * q.dC_batch_stride + (int64_t)g * q.dC_group_stride;
    const float* const x_b = static_cast<const float*>(p.x);
    const float Dd = p.D ? static_cast<const float*>(p.D)[dc] : 0.f;
    const float bias = p.delta_bias ? static_cast<const float*>(p.delta_bias)[dc] : 0.f;
    const float A_mine = static_cast<const float*>(p.A)[(int64_t)dc * p.A_d_stride + (int64_t)(NT || j < N ? j : 0) * p.A_dstate_stride];

    float dAacc = 0.f, dD_acc = 0.f, dbias_acc = 0.f;

    // B / C staging of the NEXT chunk: piece (tensor, state, j) = 8 values of one state; a thread owns PPT pieces
    RawB<T, REV> stg[PPT];
    bool st_ok = false;
    auto piece_state = [&](int h) __attribute__((always_inline)) { return (((int)threadIdx.x + W * kWave * h) >> 4) & 15; };
    auto piece_src = [&](int h) __attribute__((always_inline)) {
        const int pid = (int)threadIdx.x + W * kWave * h, ten = pid >> 8, n = NT || piece_state(h) < N ? piece_state(h) : 0;
        return ten ? Cv + (int64_t)n * p.C_dstate_stride : Bv + (int64_t)n * p.B_dstate_stride;
    };
    auto stage_issue = [&](int cc) __attribute__((always_inline)) {
        const int ll = cc * CH + j * K;
        st_ok = cc >= 0 && ll < L;
#pragma unroll
        for (int h = 0; h < PPT; ++h) stg[h].load(piece_src(h), REV ? L - ll - K : ll, st_ok && (NT || piece_state(h) < N));
    };
    auto stage_commit = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < PPT; ++h) {
            f32x4 lo, hi;
            const bool okh = st_ok && (NT || piece_state(h) < N);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                lo[i] = okh ? stg[h].at(i) : 0.f;
                hi[i] = okh ? stg[h].at(4 + i) : 0.f;
            }
            const int pid = (int)threadIdx.x + W * kWave * h;
            lds_f4* dst = bc4 + ((pid >> 4) * CH) / 4 + j;   // (tensor * N + state) * CH
            dst[0] = lo;
            dst[16] = hi;
        }
    };
    // slab reduction: output (par, f) of a pair = float f of state par; a thread owns PPT outputs and sums their W wave
    // partials.  Float f = 4 lane + k of a wave's 1 KB belongs to DPP row rho = lane >> 4: tensor rho >> 1, element
    // 4 (rho & 1) + k of position group lane & 15 (see the swaps below).  W = 8: one output, half of its partials read
    // during each state of the next pair; W = 4: two outputs (same f, par 0 / 1), one summed per state.
    // Thread -> output: consecutive threads own consecutive POSITIONS of one (tensor, state), so that a wave's atomic
    // covers 256 contiguous bytes (every atomic leaves the XCD's L2 as its own fabric request: with 16-byte runs
    // 32 bytes apart the same sums cost twice the requests -- profiles/r02_traffic_atomics.md).  Position u & 127 =
    // 8 jj + 4 rho0 + k lives in float 4 (16 (2 tensor + rho0) + jj) + k of a wave's partials; the two rho0 of a
    // wave's read share LDS banks (2-way conflict on 8 ds_read_b32 per pair: nothing).
    const int rd_par0 = (int)threadIdx.x >> 8;
    const int rd_u = threadIdx.x & 255;
    const int rd_ten = rd_u >> 7, rd_pos = rd_u & 127;
    const int rd_f = 4 * (16 * (2 * rd_ten + ((rd_pos >> 2) & 1)) + (rd_pos >> 3)) + (rd_pos & 3);
    const lds_f32* const rd_src = slab1 + rd_f;
    float* const rd_dst = rd_ten ? dCg : dBg;
    const int64_t rd_stride = rd_ten ? q.dC_dstate_stride : q.dB_dstate_stride;
    float racc = 0.f;
    float* rd_ptr = nullptr;   // state (par 0 of the pair for W = 4) this thread's pending sums go to
    bool rd_okp = false;

    RawB<T, REV> pu, pdt, pdo, pz, pout;   // row data of the NEXT chunk: requested while the current one computes
    RawB<T, REV> pdzo;                     // dz_accumulate: what dz holds (round 3: requested with the row data instead of right
                                           // before its use, where every chunk waited for it: +36 -> +? us for the accumulating launch)
    RawB<T, REV> pout2;                    // DZM == 1: the other direction's pre-gate output
    float hck_next = 0.f;
    // XL (x_has_sub == 3): the forward left the state after every 8 elements; a lane takes the one entering its elements
    // instead of rebuilding it (its own 8-step recurrence from zero + the row scan of the lane aggregates)
    float xin[XL ? kBN : 1];
    // one buffer resource per batch entry (workgroup-uniform): a batch entry's x stays under 2 GiB (host)
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(x_b) + (XL ? (int64_t)b * p.dim * p.n_chunks * p.x_chunk_stride : 0), 0,
        XL ? (int)((int64_t)p.dim * p.n_chunks * p.x_chunk_stride * 4) : 0, 0x00020000);
    const uint32_t o_xl = static_cast<uint32_t>((int64_t)dc * p.n_chunks * p.x_chunk_stride);   // the row inside its batch entry
    const int n_c = (L + CH - 1) / CH;
    const uint32_t o_x = p.x ? static_cast<uint32_t>(((int64_t)b * p.dim + dc) * p.n_chunks * p.x_chunk_stride) : 0u;
    // XL: the states entering this lane's 8 elements of chunk cc, four states (n0 .. n0 + 3) at a time: requested as soon as
    // the same registers' values for the current chunk have been used = a whole chunk ahead.  Out of range (the row's
    // first lane, chunks before the row) reads 0 through the buffer resource: no select on loaded data.
    auto request_x = [&](int cc, int n0) __attribute__((always_inline)) {
        const int idx8 = cc * (CH / 8) + j - 1;
        const uint32_t xo = cc >= 0 && idx8 >= 0 && cc * CH + j * K < L   // lanes past the row's end: 0, not unwritten memory
                                ? (o_xl + (uint32_t)((idx8 >> 8) * (int)p.x_chunk_stride + 2 * N + (idx8 & 255) * 4)) * 4u
                                : 0x80000000u;
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xo, n0 * 1024, 0));
#pragma unroll
        for (int i = 0; i < 4; ++i) xin[XL ? n0 + i : 0] = v[i];
    };
    auto request_row = [&](int cc) __attribute__((always_inline)) {
        const int ll = cc * CH + j * K;
        const bool v = cc >= 0 && ll < L && row_ok;
        const uint32_t pl = REV ? L - ll - K : ll;
        pu.load_stream(u_b, VMS_OFF(p.u_batch_stride, p.u_d_stride) + pl, v);
        pdt.load_stream(dt_b, VMS_OFF(p.delta_batch_stride, p.delta_d_stride) + pl, v);
        pdo.load_stream(dout_b, VMS_OFF(q.dout_batch_stride, q.dout_d_stride) + pl, v);
        if (HZ) {
            pz.load_stream(z_b, VMS_OFF(p.z_batch_stride, p.z_d_stride) + pl, v);
            if (DZM != 2 || out_z_b) pout.load_stream(outp_b, VMS_OFF(p.out_batch_stride, p.out_d_stride) + pl, v);
            if (DZM == 1) pout2.load_stream(out2_b, VMS_OFF(out2_batch_stride, out2_d_stride) + pl, v);
            if (DZM == 0 && q.dz_accumulate) pdzo.load_stream(dz_b, VMS_OFF(q.dz_batch_stride, q.dz_d_stride) + pl, v);
        }
        if constexpr (!XL) {
            const int e128 = cc * (CH / 128) - 1;
            const uint32_t xo = cc > 0 ? o_x + x_sub_off(e128, NT || j < N ? j : 0, (int)p.x_chunk_stride, p.x_has_sub == 3, N) : 0u;
            hck_next = x_b[xo];
        }
    };
    const int cps = (n_c + n_seg - 1) / n_seg;
    const int c_lo = seg * cps, c_hi = (c_lo + cps < n_c) ? c_lo + cps : n_c;
    float g_in = 0.f, anx_in = 1.f;
    if (seg < n_seg - 1) {
        // the carry kernel leaves one (P, q) per SUB-range (sub_m per range, scan_bwd_carry_body): chain all of them to the right
        const int n_tot = n_seg * sub_m;
        const float2* cp = seg_carry + (((int64_t)b * p.dim + dc) * n_tot) * N + j;
        for (int s2 = n_tot - 1; s2 >= (seg + 1) * sub_m; --s2) {
            const float2 pq = cp[(int64_t)s2 * N];
            g_in = fmaf(pq.x, g_in, pq.y);
        }
        const int lr = c_hi * CH;
        float t = static_cast<float>(dt_b[VMS_OFF(p.delta_batch_stride, p.delta_d_stride) + (REV ? L - 1 - lr : lr)]) + bias;
        if (p.delta_softplus) t = softplusf_(t);
        anx_in = fast_exp2(t * A_mine * kLog2e);
    }
    request_row(c_hi - 1);
    if constexpr (XL) {
#pragma unroll
        for (int n0 = 0; n0 < kBN; n0 += 4)
            if (NT || n0 < N) request_x(c_hi - 1, n0);
    }
    stage_issue(c_hi - 1);
    stage_commit();
    rec4[j] = f32x4{A_mine, c_hi > 1 ? hck_next : 0.f, anx_in, g_in};
    lds_barrier_b();
    f32x4 bc = rec4[0];
    const bool is_first = j == 0, is_last = j == 15;
    for (int c = c_hi - 1; c >= c_lo; --c) {
        const int l0 = c * CH + j * K;
        const bool ok = l0 < L && row_ok;
        const uint32_t pl0 = REV ? L - l0 - K : l0;
        const int rd_lo = c * CH + rd_pos;
        float* const rd_dst_c = rd_dst + (REV ? L - 1 - rd_lo : rd_lo);
        f2 dl2[K / 2], dlu2[K / 2], dy2[K / 2], sg2[K / 2];  // sg = d softplus / d(delta + bias), for the epilogue
        float sdl = 0.f, dl_first = 0.f;
        {
            float dy[K];
#pragma unroll
            for (int i = 0; i < K; ++i) {
                dy[i] = ok ? pdo.at(i) : 0.f;   // past the end: c = 0, a = 1 (identity for the suffix scan)
                float t = pdt.at(i) + bias, sg = 1.f;
                if (p.delta_softplus) {
                    // softplus and its derivative from ONE exp / rcp / log: e = exp(t), w = 1 + e: softplus = log1p(e) =
                    // log(w) + (e - (w - 1)) / w, sigmoid(t) = e / w -- accurate for strongly negative t, where
                    // 1 - exp(-softplus) (the previous form) cancels (selective_scan_fwd_kernel.cuh:153-156, bwd_kernel.cuh:439-452)
                    const float e = fast_exp(t);
                    const float w = 1.f + e;
                    const float rw = fast_rcp(w);
                    const float sp = fmaf(e - (w - 1.f), rw, fast_log(w));
                    sg = t <= 20.f ? e * rw : 1.f;
                    t = t <= 20.f ? sp : t;
                }
                t = ok ? t : 0.f;
                dl2[i / 2][i % 2] = t;
                sg2[i / 2][i % 2] = sg;
                sdl += t;
                if (i == 0) dl_first = t;
            }
            if (HZ) {
                float ov[K], dzv[K];
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    const float zv = pz.at(i);
                    const float s = sigmoidf_(zv);
                    const float silu = zv * s;
                    if (DZM != 2 || out_z_b) ov[i] = pout.at(i);
                    if (DZM != 2) dzv[i] = dy[i] * (DZM == 1 ? ov[i] + pout2.at(i) : ov[i]) * s * (1.f + zv * (1.f - s));
                    dy[i] *= silu;
                    if (DZM != 2 || out_z_b) ov[i] *= silu;
                }
                if (DZM == 0 && q.dz_accumulate) {  // dz += (vms_hip.h); the one-grid form of both directions writes a fresh dz
#pragma unroll
                    for (int i = 0; i < K; ++i) dzv[i] += pdzo.at(i);
                }
                if (ok) {
                    if (DZM != 2) store_b<T, REV>(dz_b + (VMS_OFF(q.dz_batch_stride, q.dz_d_stride) + pl0), dzv);
                    if (out_z_b) store_b<T, REV>(out_z_b + (VMS_OFF(p.out_z_batch_stride, p.out_z_d_stride) + pl0), ov);
                }
            }
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const float uv = pu.at(i);
                dy2[i / 2][i % 2] = dy[i];
                dlu2[i / 2][i % 2] = dl2[i / 2][i % 2] * uv;
                dD_acc = fmaf(dy[i], uv, dD_acc);
            }
        }
        f2 S1[K / 2], S2[K / 2];  // per element: sum_n g B  /  sum_n A g a x_{i-1}
#pragma unroll
        for (int k = 0; k < K / 2; ++k) {
            S1[k] = f2{0.f, 0.f};
            S2[k] = f2{0.f, 0.f};
        }
        RawB<T, REV> ukeep = pu;   // u of this chunk in its raw form, for the epilogue
        request_row(c - 1);        // in flight during the 16 states of this chunk
        stage_issue(c - 1);
#define VMS_EL(arr, i) arr[(i) / 2][(i) % 2]
        auto do_state = [&](const int n, const int i4) __attribute__((always_inline)) {
            constexpr int SG = B4<W>::kSG;
            const int par = i4 & 1;
            const int st = n % SG, buf = (n / SG) & 1;   // state inside its slab group; the group's slab buffer
            const int hh = st >> 1;                      // W = 8: which of this thread's SG / 2 outputs of the previous group
            // waves w and w + 4 share a SIMD, and between equal priorities the older wave wins every issue slot: it would
            // reach each barrier first and wait while its partner runs alone at the single-wave issue rate.  Taking turns
            // (one state each) brings both to the barrier together.
            if (W == 8 && VMS_BWD_PRIO) {
                if ((par ^ (quad >> 2)) & 1) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(0);
            }
            // fp32 B / C of this state, shared by the workgroup's 32 rows
            const lds_f4* bsrc = bc4 + (n * CH) / 4 + j;
            const f32x4 b0 = bsrc[0], b1 = bsrc[16], c0 = bsrc[kBN * CH / 4], c1 = bsrc[kBN * CH / 4 + 16];   // (the LDS layout keeps its 16-state pitch)
            // 4 wave partials of the PREVIOUS pair (other slab buffer): W = 8: half of this thread's one output;
            // W = 4: all of its output of state `par`
            float rdv[4];
            {
                // W = 8: a thread owns the outputs of states rd_par0 + 2 h (h < SG / 2) of the previous group; it reads waves 0-3 / 4-7
                // of output h during states 2 h / 2 h + 1 of this group.  W = 4: one output per state, state st of the previous group
                const lds_f32* src = rd_src + (buf ^ 1) * kB4Pair +
                                     (W == 8 ? (rd_par0 + 2 * hh) * (W * 4 * kWave) + par * 4 * (4 * kWave) : st * (W * 4 * kWave));
#pragma unroll
                for (int w = 0; w < 4; ++w) rdv[w] = src[w * (4 * kWave)];
            }
            const float Araw = bc.x, hin = bc.y, anx_n = bc.z, gin = bc.w;
            // before the last state's fetch of state 0's record, lane j stores the state entering the NEXT chunk for state j
            if (i4 == 3 && n == N - 1) rec1[4 * j + 1] = c > 1 ? hck_next : 0.f;
            bc = rec4[(n + 1) & (N - 1)];
            const float An = Araw * kLog2e;
            const f2 An2 = f2{An, An}, Araw2 = f2{Araw, Araw};
            f2 Bn2[K / 2], c2[K / 2], a2[K / 2], xs2[K / 2], ax2[K / 2];
#pragma unroll
            for (int k = 0; k < K / 2; ++k) {
                Bn2[k] = k == 0 ? f2{b0.x, b0.y} : k == 1 ? f2{b0.z, b0.w} : k == 2 ? f2{b1.x, b1.y} : f2{b1.z, b1.w};
                const f2 t = dl2[k] * An2;
                a2[k] = f2{fast_exp2(t.x), fast_exp2(t.y)};
                xs2[k] = dlu2[k] * Bn2[k];  // b_i for now
                c2[k] = (k == 0 ? f2{c0.x, c0.y} : k == 1 ? f2{c0.z, c0.w} : k == 2 ? f2{c1.x, c1.y} : f2{c1.z, c1.w}) * dy2[k];
            }
            const float a_right = bdpp<DPP_ROW_SHL1>(anx_n, a2[0].x);  // lane 15 of the row <- next chunk
            float rg = 0.f;
#pragma unroll
            for (int i = K - 1; i >= 0; --i) rg = fmaf(i == K - 1 ? a_right : VMS_EL(a2, i + 1), rg, VMS_EL(c2, i));
            float ra = fast_exp2((sdl - dl_first) * An) * a_right;
            rg = fmaf(ra, is_last ? gin : 0.f, rg);
            float xseed;
            if constexpr (XL) {
                row_scan_suffix_b(ra, rg);
                xseed = xin[n];
            } else {
                float px = 0.f;
#pragma unroll
                for (int i = 0; i < K; ++i) px = fmaf(VMS_EL(a2, i), px, VMS_EL(xs2, i));
                float pa = fast_exp2(sdl * An);
                px = fmaf(pa, is_first ? hin : 0.f, px);
                row_scan_pair_b(pa, px, ra, rg);
                xseed = bdpp<DPP_ROW_SHR1>(hin, px);
            }
            float grun = bdpp<DPP_ROW_SHL1>(gin, rg);
            if (is_first) *(lds_f2*)(rec1 + 4 * n + 2) = f2{a2[0].x, rg};
            {
                float xrun = xseed;
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    const float ax = VMS_EL(a2, i) * xrun;
                    xrun = ax + VMS_EL(xs2, i);
                    VMS_EL(ax2, i) = ax;
                    VMS_EL(xs2, i) = xrun;
                }
            }
#pragma unroll
            for (int i = K - 1; i >= 0; --i) {
                grun = fmaf(i == K - 1 ? a_right : VMS_EL(a2, i + 1), grun, VMS_EL(c2, i));
                VMS_EL(c2, i) = grun;
            }
            f2 dA2 = f2{0.f, 0.f};
            float vb[K], vc[K];   // this row's dB / dC products of the lane's 8 elements
#pragma unroll
            for (int k = 0; k < K / 2; ++k) {
                const f2 g2 = c2[k];
                const f2 gax = g2 * ax2[k];  // g * a_i x_{i-1}
                S1[k] = pk_fma_b(g2, Bn2[k], S1[k]);
                S2[k] = pk_fma_b(Araw2, gax, S2[k]);
                dA2 = pk_fma_b(dl2[k], gax, dA2);
                const f2 dBv = g2 * dlu2[k], dCv = dy2[k] * xs2[k];
                vb[2 * k] = dBv.x; vb[2 * k + 1] = dBv.y;
                vc[2 * k] = dCv.x; vc[2 * k + 1] = dCv.y;
            }
            const float dA_tot = row_allsum_b(dA2.x + dA2.y);
            if (j == n) dAacc += dA_tot;
            // rows r and r + 2: after the swap lanes 0-31 hold vb (both rows), lanes 32-63 vc
            // one asm statement per pair: with all 16 values tied to ONE statement the register allocator gathered them with ~9
            // v_mov per state (167 -> 59 per chunk, 2,511 -> 2,403 VALU instructions; the compiler's own permlane*_swap builtins
            // mis-assign the second result inside this loop -- tools/microbench/swap_probe.hip shows them correct in isolation).
            // The products are final before the first swap (no instruction moves across the scheduling barrier): its s_nop
            // covers the VALU-write -> swap-read wait states for all eight.
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(vb[0]), "+v"(vc[0]));
#pragma unroll
            for (int i = 1; i < K; ++i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(vb[i]), "+v"(vc[i]));
            f2 t2[K / 2];
#pragma unroll
            for (int k = 0; k < K / 2; ++k) t2[k] = f2{vb[2 * k], vb[2 * k + 1]} + f2{vc[2 * k], vc[2 * k + 1]};
            // elements e and e + 4: DPP row rho then holds tensor rho >> 1, elements 4 (rho & 1) + k, all 4 rows summed
            float t[K];
#pragma unroll
            for (int i = 0; i < K; ++i) t[i] = VMS_EL(t2, i);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(t[0]), "+v"(t[4]));
#pragma unroll
            for (int i = 1; i < K / 2; ++i) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(t[i]), "+v"(t[i + 4]));
            const f2 o0 = f2{t[0], t[1]} + f2{t[4], t[5]}, o1 = f2{t[2], t[3]} + f2{t[6], t[7]};
            slab4[(buf * kB4Pair + st * (W * 4 * kWave) + quad * (4 * kWave)) / 4 + lane] =
                __builtin_shufflevector(o0, o1, 0, 1, 2, 3);
            {
                const float tsum = (rdv[0] + rdv[1]) + (rdv[2] + rdv[3]);
                if (W == 8) {
                    racc = par == 0 ? tsum : racc + tsum;
                    if (par == 1 && rd_okp && !VMS_ABL_NOATOM) atomicAdd(rd_ptr + (rd_par0 + 2 * hh) * rd_stride, racc);
                } else if (rd_okp) {
                    if (!VMS_ABL_NOATOM) atomicAdd(rd_ptr + st * rd_stride, tsum);
                }
            }
            if (st == SG - 1) {
                // the group just written is summed during the next one
                rd_ptr = rd_dst_c + (int64_t)(n - (SG - 1)) * rd_stride;
                rd_okp = rd_lo < L;
                if (!VMS_ABL_NOBAR) lds_barrier_b();  // pair written by all waves; previous pair's buffer free again
            }
        };
        if constexpr (XL) {   // unrolled: xin[n] is a register
#pragma unroll
            for (int n = 0; n < kBN; n += 4) {
                if (NT || n < N) {
                    do_state(n, 0);
                    do_state(n + 1, 1);
                    do_state(n + 2, 2);
                    do_state(n + 3, 3);
                    request_x(c - 1, n);
                }
            }
        } else {
#pragma unroll 1
            for (int n = 0; n < N; n += 4) {
                do_state(n, 0);
                do_state(n + 1, 1);
                do_state(n + 2, 2);
                do_state(n + 3, 3);
            }
        }
#undef VMS_EL
        // u of this chunk before its registers are refilled; the next chunk's row data then travels during the
        // epilogue, the staging commit and the barrier
        // keep the compiler from carrying the prologue's widened u, D dy products and address arithmetic through the
        // 16 states: what the epilogue needs is rebuilt from these (opaque) registers
        asm volatile("" : "+v"(ukeep.v[0]));
#pragma unroll
        for (int k2 = 0; k2 < K / 2; ++k2) asm volatile("" : "+v"(dy2[k2]));
        float uvv[K];
#pragma unroll
        for (int i = 0; i < K; ++i) uvv[i] = ukeep.at(i);
        {
            float duv[K], ddl[K];
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const float dl = dl2[i / 2][i % 2];
                const float s1 = S1[i / 2][i % 2], s2 = S2[i / 2][i % 2];
                const float sg = sg2[i / 2][i % 2];   // sigmoid(delta_raw + bias), 1 above the reference's threshold
                duv[i] = fmaf(dl, s1, Dd * dy2[i / 2][i % 2]);
                ddl[i] = fmaf(uvv[i], s1, s2) * sg;
                dbias_acc += ok ? ddl[i] : 0.f;
            }
            if (ok) {
                store_stream_b<T, REV>(du_b + (VMS_OFF(q.du_batch_stride, q.du_d_stride) + pl0), duv);
                store_stream_b<T, REV>(ddelta_b + (VMS_OFF(q.ddelta_batch_stride, q.ddelta_d_stride) + pl0), ddl);
            }
        }
        stage_commit();   // every wave is past its last B / C read of this chunk (barrier of the last pair)
        lds_barrier_b();
    }
#undef VMS_OFF
    if (rd_okp) {   // the last group (buffer 1) is still in the slab
        constexpr int SGt = B4<W>::kSG;
#pragma unroll
        for (int h = 0; h < (W == 8 ? SGt / 2 : SGt); ++h) {
            const int par = W == 8 ? rd_par0 + 2 * h : h;
            const lds_f32* src = rd_src + kB4Pair + par * (W * 4 * kWave);
            float tsum = src[0];
#pragma unroll
            for (int w = 1; w < W; ++w) tsum += src[w * (4 * kWave)];
            atomicAdd(rd_ptr + par * rd_stride, tsum);
        }
    }
    const float dD_tot = row_allsum_b(dD_acc), db_tot = row_allsum_b(dbias_acc);
    if (row_ok) {
        if (q.dD && j == 0) atomicAdd(q.dD + d, dD_tot);
        if (q.ddelta_bias && j == 0) atomicAdd(q.ddelta_bias + d, db_tot);
        if (NT || j < N) atomicAdd(q.dA + (int64_t)d * q.dA_d_stride + (int64_t)j * q.dA_dstate_stride, dAacc);
    }
}


// RM: 0 = left-to-right, 1 = right-to-left, 2 = per batch entry (vms_hip.h reverse_from).  A workgroup serves one batch
// entry: the direction is workgroup-uniform, one branch selects the body.
// XL: x carries the forward's 8-element checkpoints (x_has_sub == 3), < 2 GiB per batch entry (one buffer resource each)
template <typename T, bool HZ, int RM, int W, bool XL, int NT = kBN>
__global__ __launch_bounds__(W* kWave, 2) void scan_bwd_pair4_kernel(const vms_scan_bwd_params q, const int n_seg, const float2* __restrict__ seg_carry,
                                                                    const int sub_m) {
    if constexpr (RM == 2) {
        const int wg_per_seg = gridDim.x / n_seg;
        const int b = (int)(blockIdx.x % wg_per_seg) % q.f.batch;
        if (b >= q.f.reverse_from) scan_bwd_pair4_body<T, HZ, true, W, XL, 0, NT>(q, n_seg, seg_carry, blockIdx.x, gridDim.x, nullptr, 0, 0, sub_m);
        else scan_bwd_pair4_body<T, HZ, false, W, XL, 0, NT>(q, n_seg, seg_carry, blockIdx.x, gridDim.x, nullptr, 0, 0, sub_m);
    } else {
        scan_bwd_pair4_body<T, HZ, RM == 1, W, XL, 0, NT>(q, n_seg, seg_carry, blockIdx.x, gridDim.x, nullptr, 0, 0, sub_m);
    }
}

// Both directions of a bidirectional block (mamba_simple.py:234-258: two parameter sets over the same rows, the second scanned
// right-to-left) in ONE grid: the first half of the workgroups runs qa left-to-right and writes the whole dz (DZM 1), the second
// half runs qb right-to-left (DZM 2).  Why: a direction of the suite's most common shape, (8, 768, 3136), is 192 workgroups of
// 8 waves for 256 CUs -- a quarter of the chip idles through both launches, and no split of ONE direction helps (the kernel
// needs its 2 waves per SIMD, profiles/r03_bwd_segments.md).  Together, as 768 workgroups of W = 4 waves (one wave per SIMD
// each, two resident per CU), the grid is 1.5 rounds whose last half-round runs one wave per SIMD at 0.7x the time:
// 2 x 317 -> 489 us (profiles/r04_dual_bwd.md).
template <typename T, int W, bool XL, int NT = kBN>
__global__ __launch_bounds__(W* kWave, 2) void scan_bwd_pair4_dual_kernel(const vms_scan_bwd_params qa, const vms_scan_bwd_params qb) {
    const int half = gridDim.x >> 1;
    if ((int)blockIdx.x < half) {
        scan_bwd_pair4_body<T, true, false, W, XL, 1, NT>(qa, 1, nullptr, blockIdx.x, half, static_cast<const T*>(qb.f.out),
                                                           qb.f.out_batch_stride, qb.f.out_d_stride);
    } else {
        scan_bwd_pair4_body<T, true, true, W, XL, 2, NT>(qb, 1, nullptr, blockIdx.x - half, half);
    }
}

// (A third generation -- state pairs in packed registers, 1,372 us against this kernel's 949 us -- was measured in round 2,
// profiles/r02_bwd_state_pairs.md, and removed from the tree in round 3; its source is csrc/selective_scan_bwd_pair.hip
// :1028-1507 of commit 64fcaa4.)

// ---- adjoint carries of a segmented backward --------------------------------------------------------------------
// With few rows and long sequences (batch 1, 768 channels, 65,536 tokens: 24 workgroups for 256 CUs) the grid
// above is repeated over n_seg ranges of chunks.  The adjoint entering a range from the right depends linearly on
// the one entering the range after it:  g_out = P g_in + q  per (row, state), with P = the product of a_{i+1}
// over the range and q = the reverse recurrence g_i = a_{i+1} g_{i+1} + C_i dy_i run from g = 0.  This kernel
// computes (P, q) for the ranges 1 .. n_seg-1 -- one exp, the C dy products, one fma chain and the reverse half of
// the row scan per (element, state), about a fifth of the full backward -- and the main kernel chains them.
template <int CTRL>
__device__ __forceinline__ float row_newbcast(float v) { return bdpp<0x150 + CTRL>(0.f, v); }

// sub_m (round 4): every range is cut into sub_m SUB-ranges of chunks with a (P, q) each -- affine maps compose, so the carry
// pass can be as parallel as the chip needs, whatever range count the main kernel's grid wants: at (1, 768, 65536) the 9 ranges x
// 192 waves left this kernel at 1.5 waves per SIMD (79 VGPRs: 6 fit) and the vector ALUs 55 % busy (profiles/r04_long_pmc.md).
template <typename T, bool HZ, bool REV>
__device__ __forceinline__ void scan_bwd_carry_body(const vms_scan_bwd_params& q, const int n_seg, float2* __restrict__ seg_carry,
                                                    const int sub_m) {
    const vms_scan_fwd_params& p = q.f;
    constexpr int K = kBK, N = kBN, CH = kCH;
    const int lane = threadIdx.x & 63;
    const int quad = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, r = lane >> 4;
    const int wg_per_sub = gridDim.x / ((n_seg - 1) * sub_m);
    const int sub = sub_m + blockIdx.x / wg_per_sub, wg = blockIdx.x % wg_per_sub;   // range 0 has nothing to its left
    const int seg = sub / sub_m, sub_k = sub - seg * sub_m;
    const int b = wg % p.batch;
    const int d0 = (wg / p.batch) * kBRows;
    const int d = d0 + quad * 4 + r;
    const bool row_ok = d < p.dim;
    const int dc = row_ok ? d : p.dim - 1;
    const int g = d0 / (p.dim / p.n_groups);
    const int L = p.seqlen;
    const T* const dt_b = static_cast<const T*>(p.delta);
    const T* const dout_b = static_cast<const T*>(q.dout);
    const T* const z_b = static_cast<const T*>(p.z);
#define VMS_OFF(bs, ds) static_cast<uint32_t>((int64_t)b * (bs) + (int64_t)dc * (ds))
    const T* const Cv = static_cast<const T*>(p.C) + (int64_t)b * p.C_batch_stride + (int64_t)g * p.C_group_stride;
    const float bias = p.delta_bias ? static_cast<const float*>(p.delta_bias)[dc] : 0.f;
    const float A_mine = static_cast<const float*>(p.A)[(int64_t)dc * p.A_d_stride + (int64_t)j * p.A_dstate_stride] * kLog2e;
    const int n_c = (L + CH - 1) / CH;
    const int cps = (n_c + n_seg - 1) / n_seg;
    const int r_lo = seg * cps, r_hi = (r_lo + cps < n_c) ? r_lo + cps : n_c;         // the range, as the main kernel cuts it
    const int cs = (cps + sub_m - 1) / sub_m;                                         // chunks per sub-range
    const int c_lo = r_lo + sub_k * cs < r_hi ? r_lo + sub_k * cs : r_hi;
    const int c_hi = c_lo + cs < r_hi ? c_lo + cs : r_hi;                            // empty (c_lo == c_hi): the identity map
    // The adjoint recurrence g_i = a_{i+1} g_{i+1} + c_i (a_k = exp2(A' delta_k), c_i = C_i dy_i) over the sub-range [lo, hi) is the affine
    // map  g_lo = P g_hi + q  with  P = exp2(A' sum_{k = lo+1 .. hi} delta_k)  and  q = sum_i exp2(A' T_i) c_i,  T_i = sum_{k = lo+1 .. i}
    // delta_k: only the AGGREGATE is wanted here, so nothing has to run sequentially -- T is a prefix sum of delta (one scan per chunk,
    // shared by the 16 states) and q a dot product: per (element, state) half a packed multiply for A' T, one exp2, half a packed
    // multiply for C dy and half a packed fma into the state's accumulator.  (Until round 4 this pass ran the recurrence itself --
    // the a_i chain, its 8 fma per state and a 4-step row scan per state and chunk: 68 instead of 28 vector instructions per state,
    // 260 us of the 920 us a long-video backward scan took.)  Exponents are <= 0 (A < 0, T >= 0): far terms underflow to 0.
    float gcar = 0.f, pacc = 1.f;              // lane j <-> state j
    const bool is_last = j == 15;
    (void)is_last;
    if (c_lo < c_hi) {
        RawB<T, REV> rcA, rcB, rdt, rdo, rz;
        f2 qacc[N];
#pragma unroll
        for (int n = 0; n < N; ++n) qacc[n] = f2{0.f, 0.f};
        float off = 0.f, dl_lo = 0.f;            // sum of delta over the chunks already walked; delta of the sub-range's first element
        for (int c = c_lo; c < c_hi; ++c) {
            const int l0 = c * CH + j * K;
            const bool ok = l0 < L && row_ok;              // seqlen % K == 0 (host): all or nothing
            const uint32_t pl0 = REV ? L - l0 - K : l0;
            const uint32_t pl0n = REV ? L - (l0 + CH) - K : l0 + CH;   // the same lane in the next chunk (c + 1)
            const bool okn = c + 1 < c_hi && l0 + CH < L && row_ok;
            if (c == c_lo) {
                rcA.load(Cv, pl0, l0 < L);
                rdt.load(dt_b, VMS_OFF(p.delta_batch_stride, p.delta_d_stride) + pl0, ok);
                rdo.load(dout_b, VMS_OFF(q.dout_batch_stride, q.dout_d_stride) + pl0, ok);
                if (HZ) rz.load(z_b, VMS_OFF(p.z_batch_stride, p.z_d_stride) + pl0, ok);
            }
            f2 T2[K / 2], dy2[K / 2];
            {
                float run = 0.f, pf[K];
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    float t = rdt.at(i) + bias;
                    if (p.delta_softplus) t = softplusf_(t);
                    t = ok ? t : 0.f;
                    float dyv = ok ? rdo.at(i) : 0.f;
                    if (HZ) {
                        const float zv = rz.at(i);
                        dyv *= zv * sigmoidf_(zv);
                    }
                    run += t;
                    pf[i] = run;                               // inclusive prefix inside the lane
                    dy2[i / 2][i % 2] = dyv;
                    if (i == 0 && c == c_lo) dl_lo = row_newbcast<0>(t);     // (uniform branch; lane 0's first element)
                }
                // exclusive scan of the lane totals over the row's 16 lanes
                float inc = run;
                inc += bdpp<DPP_ROW_SHR1>(0.f, inc);
                inc += bdpp<DPP_ROW_SHR2>(0.f, inc);
                inc += bdpp<DPP_ROW_SHR4>(0.f, inc);
                inc += bdpp<DPP_ROW_SHR8>(0.f, inc);
                const float base = off + (inc - run) - dl_lo;  // T of the element before this lane's first
                off += row_newbcast<15>(inc);
#pragma unroll
                for (int i = 0; i < K; ++i) T2[i / 2][i % 2] = base + pf[i];
            }
            {   // the next chunk's row data travels while this chunk's 16 states compute
                rdt.load(dt_b, VMS_OFF(p.delta_batch_stride, p.delta_d_stride) + pl0n, okn);
                rdo.load(dout_b, VMS_OFF(q.dout_batch_stride, q.dout_d_stride) + pl0n, okn);
                if (HZ) rz.load(z_b, VMS_OFF(p.z_batch_stride, p.z_d_stride) + pl0n, okn);
            }
#define VMS_CARRY_STATE(n, rc, rn)                                                                                  \
            {                                                                                                       \
                /* C of the next state -- after the last one: state 0 of the next chunk -- while this one computes */  \
                rn.load(Cv + (int64_t)(((n) + 1) & (N - 1)) * p.C_dstate_stride, (n) == N - 1 ? pl0n : pl0,         \
                        (n) == N - 1 ? (c + 1 < c_hi && l0 + CH < L) : l0 < L);                                      \
                const float An = row_newbcast<n>(A_mine);                                                           \
                const f2 An2 = f2{An, An};                                                                          \
                _Pragma("unroll") for (int k = 0; k < K / 2; ++k) {                                                 \
                    const f2 t = T2[k] * An2;                                                                       \
                    const f2 e2 = f2{fast_exp2(t.x), fast_exp2(t.y)};                                               \
                    const f2 c2 = f2{rc.at(2 * k), rc.at(2 * k + 1)} * dy2[k];                                      \
                    qacc[n] = pk_fma_after_trans_b(e2, c2, qacc[n]);                                                         \
                }                                                                                                   \
            }
            VMS_CARRY_STATE(0, rcA, rcB) VMS_CARRY_STATE(1, rcB, rcA) VMS_CARRY_STATE(2, rcA, rcB) VMS_CARRY_STATE(3, rcB, rcA)
            VMS_CARRY_STATE(4, rcA, rcB) VMS_CARRY_STATE(5, rcB, rcA) VMS_CARRY_STATE(6, rcA, rcB) VMS_CARRY_STATE(7, rcB, rcA)
            VMS_CARRY_STATE(8, rcA, rcB) VMS_CARRY_STATE(9, rcB, rcA) VMS_CARRY_STATE(10, rcA, rcB) VMS_CARRY_STATE(11, rcB, rcA)
            VMS_CARRY_STATE(12, rcA, rcB) VMS_CARRY_STATE(13, rcB, rcA) VMS_CARRY_STATE(14, rcA, rcB) VMS_CARRY_STATE(15, rcB, rcA)
#undef VMS_CARRY_STATE
        }
        // delta of the first element to the right of the sub-range (a_hi multiplies the entering adjoint); 0 at the sequence's end
        float dl_hi = 0.f;
        if (c_hi < n_c) {
            const int lr = c_hi * CH;
            float t = static_cast<float>(dt_b[VMS_OFF(p.delta_batch_stride, p.delta_d_stride) + (REV ? L - 1 - lr : lr)]) + bias;
            if (p.delta_softplus) t = softplusf_(t);
            dl_hi = t;
        }
        pacc = fast_exp2(A_mine * (off - dl_lo + dl_hi));
#pragma unroll
        for (int n = 0; n < N; ++n) {
            const float tot = row_allsum_b(qacc[n].x + qacc[n].y);
            if (j == n) gcar = tot;
        }
    }
    if (row_ok) seg_carry[(((int64_t)b * p.dim + d) * (n_seg * sub_m) + sub) * N + j] = float2{pacc, gcar};
#undef VMS_OFF
}

template <typename T, bool HZ, int RM>   // RM as for scan_bwd_pair4_kernel
__global__ __launch_bounds__(kBQ* kWave) void scan_bwd_carry_kernel(const vms_scan_bwd_params q, const int n_seg,
                                                                        float2* __restrict__ seg_carry, const int sub_m) {
    if constexpr (RM == 2) {
        const int wg_per_sub = gridDim.x / ((n_seg - 1) * sub_m);
        const int b = (int)(blockIdx.x % wg_per_sub) % q.f.batch;
        if (b >= q.f.reverse_from) scan_bwd_carry_body<T, HZ, true>(q, n_seg, seg_carry, sub_m);
        else scan_bwd_carry_body<T, HZ, false>(q, n_seg, seg_carry, sub_m);
    } else {
        scan_bwd_carry_body<T, HZ, RM == 1>(q, n_seg, seg_carry, sub_m);
    }
}

// reverse_from served by ONE launch (plus one carry launch when split): whole-vector rows, i.e. the second generation
bool scan_bwd_pair_native_mixed(const vms_scan_bwd_params& q) {
    const vms_scan_fwd_params& p = q.f;
    return p.reverse_from > 0 && p.reverse_from < p.batch && !p.reverse && p.seqlen % kBK == 0;
}

// how many ranges of chunks the backward is split into (1 = not split); q.f.segments >= 1 forces a count (vms_hip.h).
// Measured (profiles/r03_bwd_segments.md, one MI355X, 256 CUs; workgroups = batch x dim / 32):
//   * grids of more than half the CUs (192 workgroups: (8, 768, 3136), (8, 768, 1568)) LOSE 7 - 20 % with any split: two
//     workgroups share a CU anyway, the kernel is VALU-bound, and the carry pass is pure overhead -> never split;
//   * small grids gain up to 3.2x: 32 workgroups (one direction of the DBM block at (2, 512, 2304)) 256 -> 81 us with 6 - 8
//     ranges, 64 workgroups 259 -> 120 us with 4, 96 workgroups 360 -> 221 us with 6.
//   The best count is, with one exception, the LARGEST that still gives every workgroup its own CU (one round):
//   (1, 768, 65536) = 24 workgroups: 10 ranges 1,070 us, 9: 1,174, 12 (288 workgroups, a second round): 1,596, 16: 1,229;
//   the exception, 96 workgroups, prefers 6 (221 us) to 2 (274 us) -- both far from the unsplit 360 us.
// Rule: floor(CUs / workgroups) ranges, at least 3 chunks (384 elements) per range, at most 16.
int scan_bwd_pair_segments(const vms_scan_bwd_params& q) {
    const vms_scan_fwd_params& p = q.f;
    if (p.seqlen % kBK != 0 || p.dstate != kBN) return 1;   // (dstate 4 / 8: unsplit; the carry kernel is built for 16)
    const int n_c = (p.seqlen + kCH - 1) / kCH;
    const int n_wg = p.batch * ((p.dim + kBRows - 1) / kBRows);
    int want;
    if (p.segments >= 1) {
        want = p.segments;
    } else {
        const int cus = device_cu_count();
        want = cus / (n_wg > 0 ? n_wg : 1);
        if (want > n_c / 3) want = n_c / 3;
    }
    if (want > 16) want = 16;
    if (want > n_c) want = n_c;
    if (want < 2) return 1;
    const int cps = (n_c + want - 1) / want;
    return (n_c + cps - 1) / cps;   // no empty range
}

constexpr int kMaxSub = 4;   // sub-ranges per range of the carry pass
int64_t scan_bwd_pair_ws_bytes(const vms_scan_bwd_params& q) {
    return (int64_t)q.f.batch * q.f.dim * 16 * kMaxSub * kBN * (int64_t)sizeof(float2);   // up to 16 ranges x kMaxSub sub-ranges
}
// sub-ranges per range for the carry pass: enough 8-wave workgroups for ~6 waves per SIMD (what its 79 VGPRs admit), at least 4
// chunks per sub-range
static int scan_bwd_carry_sub(const vms_scan_bwd_params& q, int n_seg) {
    const vms_scan_fwd_params& p = q.f;
    const int n_c = (p.seqlen + kCH - 1) / kCH, cps = (n_c + n_seg - 1) / n_seg;
    const int64_t waves1 = (int64_t)p.batch * ((p.dim + kBRows - 1) / kBRows) * kBQ * (n_seg - 1);
    const int64_t want = 6 * 4 * (int64_t)device_cu_count();
    int m = (int)((want + waves1 - 1) / waves1);
    if (m > kMaxSub) m = kMaxSub;
    while (m > 1 && (cps + m - 1) / m < 4) --m;   // (4, 512, 2304): 5-chunk ranges cut in 2-chunk pieces cost more than they hide
    return m < 1 ? 1 : m;
}

// would scan_bwd_pair4_kernel<.., XL = true> serve the backward of this forward?  (the shape conditions of
// scan_bwd_pair_eligible + whole-vector rows + one buffer resource over x at the 258 * dstate pitch)
bool scan_bwd_pair_lane_ckpt_ok(const vms_scan_fwd_params& p) {
    if (!p.is_variable_B || !p.is_variable_C || (p.dstate != kBN && p.dstate != 8 && p.dstate != 4) || p.n_groups < 1 || p.dim % p.n_groups != 0) return false;
    if ((p.dim / p.n_groups) % kBRows != 0) return false;
    // seqlen % 16 == 0: the forward's LDS kernel writes the checkpoints as whole lines; the per-wave kernel that serves other
    // lengths writes them 4 bytes at a time ((8, 768, 3144): forward 167 -> 226 us for a backward 396 -> 359 us)
    if (p.seqlen % 16 != 0) return false;
    const int64_t n_chunks = (p.seqlen + 2047) / 2048, lim = (int64_t)1 << 31;
    // a batch entry's x under 2 GiB (one buffer resource per batch entry), all of x under 2^31 elements (scan_bwd_pair_eligible)
    return (int64_t)p.dim * n_chunks * 258 * p.dstate * 4 < lim && (int64_t)p.batch * p.dim * n_chunks * 258 * p.dstate < lim;
}

bool scan_bwd_pair_eligible(const vms_scan_bwd_params& q, bool vec) {
    const vms_scan_fwd_params& p = q.f;
    (void)vec;  // 16-byte vector accesses need no alignment on gfx950
    if (!p.is_variable_B || !p.is_variable_C || (p.dstate != kBN && p.dstate != 8 && p.dstate != 4) || !p.x || (p.x_has_sub != 1 && p.x_has_sub != 3)) return false;
    const int dpg = p.dim / p.n_groups;
    if (dpg % kBRows != 0) return false;     // a workgroup's rows must share one B/C group
    if (p.dstate != kBN && p.seqlen % kBK != 0) return false;   // dstate 4 / 8 (round 6): the whole-vector kernel only
    if (p.seqlen % kBK != 0 && p.bc_pad < kBK - p.seqlen % kBK) return false;   // ragged: B / C padding needed
    // 32-bit element offsets inside the kernel
    const int64_t lim = (int64_t)1 << 31;
    auto span = [&](int64_t bs, int64_t ds) { return (p.batch - 1) * bs + (p.dim - 1) * ds + p.seqlen; };
    if (span(p.u_batch_stride, p.u_d_stride) >= lim || span(p.delta_batch_stride, p.delta_d_stride) >= lim ||
        span(q.dout_batch_stride, q.dout_d_stride) >= lim || span(q.du_batch_stride, q.du_d_stride) >= lim ||
        span(q.ddelta_batch_stride, q.ddelta_d_stride) >= lim || span(p.z_batch_stride, p.z_d_stride) >= lim ||
        span(p.out_batch_stride, p.out_d_stride) >= lim || span(q.dz_batch_stride, q.dz_d_stride) >= lim ||
        span(p.out_z_batch_stride, p.out_z_d_stride) >= lim ||
        (int64_t)p.batch * p.dim * p.n_chunks * p.x_chunk_stride >= lim ||
        (int64_t)(p.dstate - 1) * p.B_dstate_stride + p.seqlen >= lim || (int64_t)(p.dstate - 1) * p.C_dstate_stride + p.seqlen >= lim)
        return false;
    return true;
}

template <typename T>
static int launch_bpair(const vms_scan_bwd_params& q, hipStream_t stream) {
    const vms_scan_fwd_params& p = q.f;
    const int tiles = (p.dim + kBRows - 1) / kBRows;
    dim3 grid(p.batch * tiles), block(kBQ * kWave);
    const size_t smem = sizeof(float) * (kBcFloats + 2 * kSlabFloats + kBRows * kBN * 4);  // 16 KB + 2 x 64 KB + 8 KB
    // more than the default 64 KB of LDS per workgroup: admitted per kernel AND per device, before the first launch there
    static PerDeviceOnce attr_once;
    const hipError_t arc = attr_once.run([&]() -> hipError_t {
        hipError_t e = hipSuccess;
#define VMS_A2(Z_, R_, G_)                                                                                               \
        if (e == hipSuccess)                                                                                             \
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&scan_bwd_pair_kernel<T, Z_, R_, G_>),                    \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)
#define VMS_A(Z_, R_) VMS_A2(Z_, R_, true); VMS_A2(Z_, R_, false)
        VMS_A(true, true); VMS_A(true, false); VMS_A(false, true); VMS_A(false, false);
#undef VMS_A
#undef VMS_A2
        return e;
    });
    if (arc != hipSuccess) {
        set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d) failed: %s", (int)smem, hipGetErrorString(arc));
        return VMS_ERR_LAUNCH;
    }
    const bool rag = p.seqlen % kBK != 0;
    const bool mixed = scan_bwd_pair_native_mixed(q);   // reverse_from: the host splits everything else into two problems
    int n_seg = 1;
    float2* carry = nullptr;
    if (p.workspace != nullptr && p.workspace_bytes >= scan_bwd_pair_ws_bytes(q)) {
        n_seg = scan_bwd_pair_segments(q);
        carry = static_cast<float2*>(p.workspace);
    }
    int sub_m = 1;
    if (n_seg > 1) {
        sub_m = scan_bwd_carry_sub(q, n_seg);
        dim3 cgrid(p.batch * tiles * (n_seg - 1) * sub_m);
#define VMS_C(Z_, R_) hipLaunchKernelGGL((scan_bwd_carry_kernel<T, Z_, R_>), cgrid, block, 0, stream, q, n_seg, carry, sub_m)
        if (mixed) { if (p.z) VMS_C(true, 2); else VMS_C(false, 2); }
        else if (p.reverse) { if (p.z) VMS_C(true, 1); else VMS_C(false, 1); }
        else { if (p.z) VMS_C(true, 0); else VMS_C(false, 0); }
#undef VMS_C
        grid = dim3(p.batch * tiles * n_seg);
    }
    // whole-vector rows run on the second generation (scan_bwd_pair4_kernel, W waves per workgroup)
    constexpr bool four = true;   // whole-vector rows: second generation, 8 waves (32 rows) per workgroup
#ifndef VMS_BWD_WK
#define VMS_BWD_WK 8
#endif
    constexpr int WK = VMS_BWD_WK;
    const size_t smem4 = B4<WK>::kSmem;
    if (four && smem4 > 64 * 1024) {   // 4-state slab groups: 88 KB
        static PerDeviceOnce attr4_once;
        const hipError_t arc4 = attr4_once.run([&]() -> hipError_t {
            hipError_t e = hipSuccess;
#define VMS_A4X(Z_, R_, X_)                                                                                              \
            if (e == hipSuccess)                                                                                         \
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(&scan_bwd_pair4_kernel<T, Z_, R_, WK, X_>),         \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem4)
#define VMS_A4(Z_, R_) VMS_A4X(Z_, R_, false); VMS_A4X(Z_, R_, true)
            VMS_A4(true, 0); VMS_A4(true, 1); VMS_A4(true, 2); VMS_A4(false, 0); VMS_A4(false, 1); VMS_A4(false, 2);
#undef VMS_A4
#undef VMS_A4X
            return e;
        });
        if (arc4 != hipSuccess) {
            set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d) failed: %s", (int)smem4, hipGetErrorString(arc4));
            return VMS_ERR_LAUNCH;
        }
    }
    const dim3 grid4(p.batch * ((p.dim + 4 * WK - 1) / (4 * WK)) * (grid.x / (p.batch * tiles))), block4(WK * kWave);
    // the forward's 8-element checkpoints (x_has_sub == 3), addressed through one buffer resource
    const bool xl = p.x_has_sub == 3 && (int64_t)p.dim * p.n_chunks * p.x_chunk_stride * 4 < ((int64_t)1 << 31);
    // dstate 4 / 8: the run-time-dstate instantiation, 4-wave workgroups of 16 rows, never split
    const bool small_n = p.dstate != kBN;
    const dim3 grid_n(p.batch * ((p.dim + 15) / 16)), block_n(4 * kWave);
    if (small_n && B4<4>::kSmem > 64 * 1024) {
        static PerDeviceOnce attrn_once;
        const hipError_t arcn = attrn_once.run([&]() -> hipError_t {
            hipError_t e = hipSuccess;
#define VMS_ANX(Z_, R_, X_)                                                                                              \
            if (e == hipSuccess)                                                                                         \
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(&scan_bwd_pair4_kernel<T, Z_, R_, 4, X_, 0>),       \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)B4<4>::kSmem)
#define VMS_AN(Z_, R_) VMS_ANX(Z_, R_, false); VMS_ANX(Z_, R_, true)
            VMS_AN(true, 0); VMS_AN(true, 1); VMS_AN(true, 2); VMS_AN(false, 0); VMS_AN(false, 1); VMS_AN(false, 2);
#undef VMS_AN
#undef VMS_ANX
            return e;
        });
        if (arcn != hipSuccess) {
            set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d) failed: %s", (int)B4<4>::kSmem, hipGetErrorString(arcn));
            return VMS_ERR_LAUNCH;
        }
    }
#define VMS_L4(Z_, R_)                                                                                             \
    do {                                                                                                           \
        if (small_n) {                                                                                             \
            if (xl) hipLaunchKernelGGL((scan_bwd_pair4_kernel<T, Z_, R_, 4, true, 0>), grid_n, block_n, B4<4>::kSmem, stream, q, 1, carry, 1); \
            else hipLaunchKernelGGL((scan_bwd_pair4_kernel<T, Z_, R_, 4, false, 0>), grid_n, block_n, B4<4>::kSmem, stream, q, 1, carry, 1);   \
        } else if (xl) hipLaunchKernelGGL((scan_bwd_pair4_kernel<T, Z_, R_, WK, true>), grid4, block4, smem4, stream, q, n_seg, carry, sub_m); \
        else hipLaunchKernelGGL((scan_bwd_pair4_kernel<T, Z_, R_, WK, false>), grid4, block4, smem4, stream, q, n_seg, carry, sub_m);   \
    } while (0)
#define VMS_L(Z_, R_)                                                                                              \
    do {                                                                                                           \
        if (rag) hipLaunchKernelGGL((scan_bwd_pair_kernel<T, Z_, R_, true>), grid, block, smem, stream, q, 1, carry); \
        else if (four) VMS_L4(Z_, R_ ? 1 : 0);                                                                     \
        else hipLaunchKernelGGL((scan_bwd_pair_kernel<T, Z_, R_, false>), grid, block, smem, stream, q, n_seg, carry); \
    } while (0)
    if (mixed) {
        if (p.z) VMS_L4(true, 2);
        else VMS_L4(false, 2);
    } else if (p.reverse) { if (p.z) VMS_L(true, true); else VMS_L(false, true); }
    else { if (p.z) VMS_L(true, false); else VMS_L(false, false); }
#undef VMS_L
#undef VMS_L4
    VMS_LAUNCH_CHECK();
    set_last_kernel(small_n ? (mixed ? "scan_bwd_pair4_n+mixed" : "scan_bwd_pair4_n")
                        : mixed ? (n_seg > 1 ? "scan_bwd_pair4+mixed+split" : "scan_bwd_pair4+mixed") : rag ? "scan_bwd_pair_ragged"
                        : four ? (n_seg > 1 ? "scan_bwd_pair4+split" : "scan_bwd_pair4")
                               : (n_seg > 1 ? "scan_bwd_pair+split" : "scan_bwd_pair"));
    return VMS_OK;
}

// ---- both directions of a bidirectional block as one grid (vms_selective_scan_bwd_dual) ------------------------------------
// Conditions: the two problems are the two directions of one block -- same sizes and dtype, a left-to-right, b right-to-left,
// the SAME z and dout (then dz = dout (out_a + out_b) dsilu(z) is what z receives through both), whole-vector rows, the same
// checkpoint layout -- and together they fill the chip (below that each direction is better off split into ranges of chunks on
// its own, scan_bwd_pair_segments).
bool scan_bwd_pair_dual_fusable(const vms_scan_bwd_params& a, const vms_scan_bwd_params& b) {
    const vms_scan_fwd_params &pa = a.f, &pb = b.f;
    if (pa.batch != pb.batch || pa.dim != pb.dim || pa.seqlen != pb.seqlen || pa.dtype != pb.dtype || pa.n_groups != pb.n_groups ||
        pa.dstate != pb.dstate)
        return false;
    if (pa.reverse != 0 || pb.reverse == 0 || pa.reverse_from != 0 || pb.reverse_from != 0) return false;
    if (pa.dtype == VMS_F32) return false;   // 16-bit activations (what autocast runs); the fp32 body has no registers to spare
    if (pa.seqlen % kBK != 0 || pa.segments > 1 || pb.segments > 1) return false;
    if (!pa.z || pa.z != pb.z || pa.z_batch_stride != pb.z_batch_stride || pa.z_d_stride != pb.z_d_stride) return false;
    if (a.dout != b.dout || a.dout_batch_stride != b.dout_batch_stride || a.dout_d_stride != b.dout_d_stride) return false;
    if (!a.dz || a.dz_accumulate || !pa.out || !pb.out || pa.out == pb.out) return false;
    if (!scan_bwd_pair_eligible(a, true) || !scan_bwd_pair_eligible(b, true)) return false;
    auto xl = [](const vms_scan_fwd_params& p) {
        return p.x_has_sub == 3 && (int64_t)p.dim * p.n_chunks * p.x_chunk_stride * 4 < ((int64_t)1 << 31);
    };
    if (xl(pa) != xl(pb)) return false;
    const int n8 = pa.batch * ((pa.dim + kBRows - 1) / kBRows);
    return 2 * n8 >= device_cu_count();
}

template <typename T>
static int launch_bpair_dual(const vms_scan_bwd_params& a, const vms_scan_bwd_params& b, hipStream_t stream) {
    const vms_scan_fwd_params& p = a.f;
    const int n8 = p.batch * ((p.dim + kBRows - 1) / kBRows), cus = device_cu_count();
    // whole rounds of 8-wave workgroups at 2 waves per SIMD (the benchmark shape: 512 workgroups = 2 per CU) are the better
    // kernel (793 vs 835 us per direction there); everything else runs 4-wave workgroups, whose partial last round has one
    // wave per SIMD (profiles/r04_dual_bwd.md)
    const bool small_n = p.dstate != kBN;                  // dstate 4 / 8: the run-time-dstate instantiation (4-wave workgroups)
#ifndef VMS_DUAL_FORCE_W4
#define VMS_DUAL_FORCE_W4 0   /* 1 (A/B builds): 4-wave workgroups also for whole rounds of 8-wave ones */
#endif
    const bool w8 = (2 * n8) % cus == 0 && !small_n && !VMS_DUAL_FORCE_W4;
    const bool xl = p.x_has_sub == 3;
    if (B4<8>::kSmem > 64 * 1024 || B4<4>::kSmem > 64 * 1024) {
        static PerDeviceOnce attrd_once;
        const hipError_t arc = attrd_once.run([&]() -> hipError_t {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&scan_bwd_pair4_dual_kernel<T, 8, true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)B4<8>::kSmem);
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(&scan_bwd_pair4_dual_kernel<T, 8, false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)B4<8>::kSmem);
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(&scan_bwd_pair4_dual_kernel<T, 4, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)B4<4>::kSmem);
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(&scan_bwd_pair4_dual_kernel<T, 4, false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)B4<4>::kSmem);
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(&scan_bwd_pair4_dual_kernel<T, 4, true, 0>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)B4<4>::kSmem);
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(&scan_bwd_pair4_dual_kernel<T, 4, false, 0>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)B4<4>::kSmem);
            return e;
        });
        if (arc != hipSuccess) {
            set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed: %s", hipGetErrorString(arc));
            return VMS_ERR_LAUNCH;
        }
    }
    if (w8) {
        const dim3 grid(2 * n8), block(8 * kWave);
        if (xl) hipLaunchKernelGGL((scan_bwd_pair4_dual_kernel<T, 8, true>), grid, block, B4<8>::kSmem, stream, a, b);
        else hipLaunchKernelGGL((scan_bwd_pair4_dual_kernel<T, 8, false>), grid, block, B4<8>::kSmem, stream, a, b);
    } else if (small_n) {
        const dim3 grid(2 * p.batch * ((p.dim + 15) / 16)), block(4 * kWave);
        if (xl) hipLaunchKernelGGL((scan_bwd_pair4_dual_kernel<T, 4, true, 0>), grid, block, B4<4>::kSmem, stream, a, b);
        else hipLaunchKernelGGL((scan_bwd_pair4_dual_kernel<T, 4, false, 0>), grid, block, B4<4>::kSmem, stream, a, b);
    } else {
        const dim3 grid(2 * p.batch * ((p.dim + 15) / 16)), block(4 * kWave);
        if (xl) hipLaunchKernelGGL((scan_bwd_pair4_dual_kernel<T, 4, true>), grid, block, B4<4>::kSmem, stream, a, b);
        else hipLaunchKernelGGL((scan_bwd_pair4_dual_kernel<T, 4, false>), grid, block, B4<4>::kSmem, stream, a, b);
    }
    VMS_LAUNCH_CHECK();
    set_last_kernel(w8 ? "scan_bwd_pair4_dual_w8" : small_n ? "scan_bwd_pair4_dual_n" : "scan_bwd_pair4_dual_w4");
    return VMS_OK;
}

int launch_scan_bwd_pair_dual(const vms_scan_bwd_params& a, const vms_scan_bwd_params& b, hipStream_t stream) {
    switch (a.f.dtype) {
        case VMS_BF16: return launch_bpair_dual<bf16_t>(a, b, stream);
        case VMS_F16: return launch_bpair_dual<f16_t>(a, b, stream);
        default: set_error("dual backward: 16-bit activations only"); return VMS_ERR_UNSUPPORTED;
    }
}

int launch_scan_bwd_pair(const vms_scan_bwd_params& q, hipStream_t stream) {
    switch (q.f.dtype) {
        case VMS_BF16: return launch_bpair<bf16_t>(q, stream);
        case VMS_F16: return launch_bpair<f16_t>(q, stream);
        default: return launch_bpair<float>(q, stream);
    }
}

}  // namespace vms
