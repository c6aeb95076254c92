// selective_scan_bwd_occ4.hip -- backward selective scan of BOTH directions of a bidirectional block in <= 128 VGPRs:
// 4 resident waves per SIMD instead of 2 (round 5; replaces selective_scan_bwd_kernel, mamba/csrc/selective_scan/
// selective_scan_bwd_kernel.cuh:75-489, whose own launch bounds make the same occupancy-for-registers trade, :44-47).
//
// Why: scan_bwd_pair4_body (selective_scan_bwd_pair.hip) needs 226-256 VGPRs = 2 waves per SIMD, and its run time is its VALU
// time at the 2-waves-per-SIMD price list (profiles/r05_microbench_mix.txt: v_pk_* 5.3, DPP 5.3, fp32 VALU 3.05, v_exp_f32 /
// v_permlane*_swap 8.8 cycles per wave-instruction; at 4 waves per SIMD 4.55 / 4.64 / 2.6 / 8.4).  The benchmark shape's dual launch
// is 4,096 waves = exactly 4 per SIMD, so a body that fits 128 registers runs the whole grid resident at once.
//
// Same decomposition and arithmetic as scan_bwd_pair4_body (a wave = 4 rows, lane = 16 r + j owns 8 consecutive elements of a
// 128-element chunk, chunks walked from the end, element-pair packed math, the forward's 8-element checkpoints as seeds, B / C
// as fp32 in LDS, dB / dC summed over the workgroup's rows before one atomic per 32 rows).  What changed to get there:
//   * the adjoint recurrence runs FIRST (it needs a and C dy only), g replaces C dy in place; the state recurrence then streams
//     element pair by element pair with everything that consumes x_i / a_i x_{i-1} (S1, S2, dA, dB, dC) issued on the spot: no
//     8-element arrays of x and a x_{i-1};
//   * S2 accumulates A log2(e) g a x_{i-1} (the same register pair that scales the exp2 arguments) and is scaled by ln 2 once
//     per chunk: no second broadcast pair of A;
//   * the next chunk's row data is requested BEHIND the state loop, into the registers the states just vacated, the forward's
//     checkpoints one state group (4 states) ahead instead of a chunk ahead (8 instead of 16 registers);
//   * sigmoid(delta_raw + bias) travels from the prologue to the epilogue as fp16 pairs (4 registers; it multiplies a result
//     that is rounded to 16 bits anyway: 16-bit activations only);
//   * the 4 rows of a wave are summed through 4 KB of wave-private LDS (four ds_write_b128 + four ds_read_b128) instead of 12
//     v_permlane32/16_swap: -100 cycles of VALU time per state; the LDS round trip that made this a loss at 2 waves per SIMD
//     (profiles/r05_bwd_lds_tr.md) hides behind the other three waves;
//   * the workgroup-level sums run one state behind (two 1-state slab buffers, a barrier per state) so that LDS stays within
//     72.7 KB per 8-wave workgroup = two workgroups per CU.
// Serves vms_selective_scan_bwd_dual for 16-bit activations with the forward's 8-element checkpoints (x_has_sub == 3); every
// other case stays on selective_scan_bwd_pair.hip.
#include "vms_common.h"
#include "scan_bwd_helpers.h"
#include <type_traits>
#include <stdlib.h>

namespace vms {

typedef _Float16 h2_t __attribute__((ext_vector_type(2)));

template <int W> struct O4 {
    static constexpr int kRows = 4 * W;
    static constexpr int kSlab = W * 4 * kWave;              // floats of one state's wave partials: [wave][4 lane + k]
    static constexpr int kPPT = 8 / W;                       // B / C pieces per thread and chunk
    static constexpr int kRecPitch = kBN + 1;                // records of 16 bytes per row + one of padding (bank spread)
    static constexpr int kRec = kRows * kRecPitch * 4;
    static constexpr int kTr = W * 4 * 4 * kWave;            // [wave][piece = (tensor, element half)][lane] float4
    static constexpr int kRowc = kRows * 4;                  // per row: {D, delta_bias, running dD, running ddelta_bias}
    static constexpr size_t kSmem = sizeof(float) * (kBcFloats + 2 * kSlab + kRec + kTr + kRowc);
};

// DZM 1: this direction's workgroups write dz = dout (out + out2) dsilu(z) (the gradient z receives through both directions);
// DZM 2: no dz.  bid: the workgroup's index inside its direction.
template <typename T, bool REV, int W, int DZM>
__device__ __forceinline__ void scan_bwd_o4_body(const vms_scan_bwd_params& q, const int bid, const T* __restrict__ out2_b,
                                                 const int64_t out2_batch_stride, const int64_t out2_d_stride,
                                                 unsigned long long* __restrict__ prof = nullptr) {
#ifdef VMS_O4_PROF   /* diagnostic builds: shader-clock time per phase and wave (tools/o4_phases.py) */
    unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, plast = __builtin_readcyclecounter();
#define VMS_PT(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); pt[k] += t_ - plast; plast = t_; } while (0)
#else
#define VMS_PT(k) do { } while (0)
#endif
    static_assert(sizeof(T) == 2, "16-bit activations");
    const vms_scan_fwd_params& p = q.f;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int K = kBK, N = kBN, CH = kCH;
    constexpr int kSlab = O4<W>::kSlab, PPT = O4<W>::kPPT, kRows = O4<W>::kRows;
    lds_f4* const bc4 = (lds_f4*)smem;                                    // fp32 B / C of the chunk: [tensor][state][128]
    lds_f4* const slab4 = (lds_f4*)(smem + kBcFloats);                    // [buf][wave][lane] float4
    const lds_f32* const slab1 = (const lds_f32*)(smem + kBcFloats);
    const int lane = threadIdx.x & 63;
    const int quad = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, r = lane >> 4;
    lds_f4* const rec4 = (lds_f4*)(smem + kBcFloats + 2 * kSlab) + (quad * 4 + r) * O4<W>::kRecPitch;
    lds_f32* const rec1 = (lds_f32*)rec4;
    lds_f4* const tr4 = (lds_f4*)(smem + kBcFloats + 2 * kSlab + O4<W>::kRec) + quad * (4 * kWave);
    const int b = bid % p.batch;
    const int d0 = (bid / p.batch) * kRows;
    const int d = d0 + quad * 4 + r;
    const bool row_ok = d < p.dim;
    const int dc = row_ok ? d : p.dim - 1;
    const int g = d0 / (p.dim / p.n_groups);  // host guarantees one group per workgroup
    const int L = p.seqlen;

    // Every global access goes through a buffer resource (base in SGPRs) with the workgroup- and wave-uniform part of the
    // offset -- batch entry, the wave's first row -- in the SGPR operand and only (row inside the wave) x stride + position in a
    // 32-bit VGPR that is rebuilt per chunk: no 64-bit per-lane pointers live across the state loop (the first build of this
    // body spilled 50 registers per chunk for them: 4.6 GB of scratch traffic per launch).  Lanes that must not touch memory get
    // the offset 0x80000000 = out of range: loads return 0, stores are dropped.  The host admits only tensors whose extents keep
    // the uniform part under 4 GiB and the lane part under 2 GiB (scan_bwd_o4_dual_takes).
    typedef uint32_t u32x4_o __attribute__((ext_vector_type(4)));
    auto mk = [](const void* ptr) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, 0x7ffffffe, 0x00020000);
    };
    const int dw = d0 + quad * 4;   // the wave's first row (uniform)
    const int dwc = dw < p.dim ? dw : p.dim - 1;
#define VMS_SOFF(bs, ds) static_cast<uint32_t>(((int64_t)b * (bs) + (int64_t)dwc * (ds)) * (int64_t)sizeof(T))
    const __amdgpu_buffer_rsrc_t rs_u = mk(p.u), rs_dt = mk(p.delta), rs_do = mk(q.dout), rs_z = mk(p.z), rs_out = mk(p.out), rs_out2 = mk(out2_b),
                                 rs_du = mk(q.du), rs_ddt = mk(q.ddelta), rs_dz = mk(q.dz);
    const uint32_t so_u = VMS_SOFF(p.u_batch_stride, p.u_d_stride), so_dt = VMS_SOFF(p.delta_batch_stride, p.delta_d_stride),
                   so_do = VMS_SOFF(q.dout_batch_stride, q.dout_d_stride), so_z = VMS_SOFF(p.z_batch_stride, p.z_d_stride),
                   so_out = VMS_SOFF(p.out_batch_stride, p.out_d_stride), so_out2 = VMS_SOFF(out2_batch_stride, out2_d_stride),
                   so_du = VMS_SOFF(q.du_batch_stride, q.du_d_stride), so_ddt = VMS_SOFF(q.ddelta_batch_stride, q.ddelta_d_stride),
                   so_dz = VMS_SOFF(q.dz_batch_stride, q.dz_d_stride);
#undef VMS_SOFF
    // lane part of a row tensor's byte offset: (row inside the wave) * d stride + position.  `rr` is made opaque once per chunk
    // (asm below) so that the nine products are rebuilt there (one v_mad each) instead of living in -- or spilling from -- nine
    // registers across the state loop.
    int rr = r;
    auto voff = [&](int64_t ds, uint32_t pl, bool valid) __attribute__((always_inline)) -> uint32_t {
        const uint32_t o = (uint32_t)rr * (uint32_t)(ds * (int64_t)sizeof(T)) + pl * (uint32_t)sizeof(T);
        return valid ? o : 0x80000000u;
    };
    auto ld8 = [&](RawB<T, REV>& dst, __amdgpu_buffer_rsrc_t rs, uint32_t vo, uint32_t so) __attribute__((always_inline)) {
        dst.v[0] = __builtin_bit_cast(vec_t<T, 8>, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 2));   // nt: touched once
    };
    auto st8 = [&](const float (&in)[kBK], __amdgpu_buffer_rsrc_t rs, uint32_t vo, uint32_t so, auto aux_tag) __attribute__((always_inline)) {
        constexpr int aux = decltype(aux_tag)::value;
        vec_t<T, 8> t;
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = static_cast<T>(in[REV ? 7 - e : e]);
        const u32x4_o tv = __builtin_bit_cast(u32x4_o, t);
        __builtin_amdgcn_raw_buffer_store_b128(tv, rs, vo, so, aux);
        // gfx950: a 16-byte store with an SGPR offset reads its data registers up to two wait states after issue, and hipcc's
        // hazard recogniser takes the hazard to be absent with an SGPR offset: a VALU write of the first data register right
        // behind the store reached memory now and then (0.01 % of the vectors: du / ddelta with their first two elements 0).
        // The data stays live across two wait states (selective_scan_fwd_pair.hip flush_park).
        asm volatile("s_nop 1" ::"v"(tv));
    };
    const __amdgpu_buffer_rsrc_t rs_B = mk(static_cast<const T*>(p.B) + (int64_t)b * p.B_batch_stride + (int64_t)g * p.B_group_stride),
                                 rs_C = mk(static_cast<const T*>(p.C) + (int64_t)b * p.C_batch_stride + (int64_t)g * p.C_group_stride);
    const __amdgpu_buffer_rsrc_t rs_dB = mk(q.dB + (int64_t)b * q.dB_batch_stride + (int64_t)g * q.dB_group_stride),
                                 rs_dC = mk(q.dC + (int64_t)b * q.dC_batch_stride + (int64_t)g * q.dC_group_stride);
    // what a row needs once per chunk lives in LDS, not in registers that would cross the state loop: {D, delta_bias, running
    // sum of dy u, running sum of ddelta} per row (the lanes of a row read the same address; lane j == 0 updates the sums)
    lds_f4* const rowc = (lds_f4*)(smem + kBcFloats + 2 * kSlab + O4<W>::kRec + O4<W>::kTr) + quad * 4 + r;
    if (j == 0)
        *rowc = f32x4{p.D ? static_cast<const float*>(p.D)[dc] : 0.f, p.delta_bias ? static_cast<const float*>(p.delta_bias)[dc] : 0.f, 0.f, 0.f};

    float dAacc = 0.f;

    // ---- B / C staging of the next chunk: piece (tensor, state, j) = 8 values of one state; a thread owns PPT pieces ----
    RawB<T, REV> stg[PPT];
    bool st_ok = false;
    auto stage_issue = [&](int cc) __attribute__((always_inline)) {
        const int ll = cc * CH + j * K;
        st_ok = cc >= 0 && ll < L;
        const uint32_t pl = (uint32_t)(REV ? L - ll - K : ll) * (uint32_t)sizeof(T);
#pragma unroll
        for (int h = 0; h < PPT; ++h) {
            const int pid = (int)threadIdx.x + W * kWave * h, n = (pid >> 4) & 15;
            const bool ten = __builtin_amdgcn_readfirstlane(pid >> 8) != 0;   // a wave's pieces belong to one tensor
            const uint32_t o = (uint32_t)n * (uint32_t)((ten ? p.C_dstate_stride : p.B_dstate_stride) * (int64_t)sizeof(T)) + pl;
            const uint32_t vo = st_ok ? o : 0x80000000u;
            if (ten) stg[h].v[0] = __builtin_bit_cast(vec_t<T, 8>, __builtin_amdgcn_raw_buffer_load_b128(rs_C, vo, 0, 0));
            else stg[h].v[0] = __builtin_bit_cast(vec_t<T, 8>, __builtin_amdgcn_raw_buffer_load_b128(rs_B, vo, 0, 0));
        }
    };
    auto stage_commit = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < PPT; ++h) {
            f32x4 lo, hi;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                lo[i] = st_ok ? stg[h].at(i) : 0.f;
                hi[i] = st_ok ? stg[h].at(4 + i) : 0.f;
            }
            const int pid = (int)threadIdx.x + W * kWave * h;
            lds_f4* dst = bc4 + ((pid >> 4) * CH) / 4 + j;   // (tensor * N + state) * CH
            dst[0] = lo;
            dst[16] = hi;
        }
    };
    // ---- workgroup-level sum of the waves' partials, one state behind: output u of a state = float rd_f of every wave's
    // 1 KB (DPP row rho = lane >> 4 holds tensor rho >> 1, elements 4 (rho & 1) + k of position group lane & 15); consecutive
    // threads own consecutive POSITIONS so that a wave's atomic covers 256 contiguous bytes.  W = 8: the two halves of the
    // workgroup take turns (even states: waves 0-3); W = 4: every thread owns one output of every state.
    const int rd_u = threadIdx.x & 255;
    const int rd_ten = rd_u >> 7, rd_pos = rd_u & 127;
    const int rd_f = 4 * (16 * (2 * rd_ten + ((rd_pos >> 2) & 1)) + (rd_pos >> 3)) + (rd_pos & 3);
    const lds_f32* const rd_src = slab1 + rd_f;
    const int rd_turn = W == 8 ? (int)(threadIdx.x >> 8) : 0;   // parity of the states this thread sums (W == 8)
    const bool rd_tenu = __builtin_amdgcn_readfirstlane(rd_ten) != 0;   // a wave's 64 outputs belong to one tensor
    const uint32_t rd_sstride = (uint32_t)((rd_tenu ? q.dC_dstate_stride : q.dB_dstate_stride) * 4);
    uint32_t rd_vo = 0x80000000u;   // byte offset of this thread's position in the chunk whose sums are pending (out of range: none)
    uint32_t rd_so = 0;             // + the pending state's row

    RawB<T, REV> pu, pdt, pdo, pz, pout, pout2;   // row data of the next chunk
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(static_cast<const float*>(p.x)) + (int64_t)b * p.dim * p.n_chunks * p.x_chunk_stride, 0,
        (int)((int64_t)p.dim * p.n_chunks * p.x_chunk_stride * 4), 0x00020000);
    const uint32_t o_xl = static_cast<uint32_t>((int64_t)dc * p.n_chunks * p.x_chunk_stride);   // the row inside its batch entry
    const int n_c = (L + CH - 1) / CH;
    // the states entering this lane's 8 elements of chunk cc, states n0 .. n0 + 3 (the forward's 8-element checkpoints);
    // out of range (the row's first lane, chunks before the row, lanes past the end) reads 0 through the buffer resource
    auto request_x = [&](int cc, int n0) __attribute__((always_inline)) -> f32x4 {
        const int idx8 = cc * (CH / 8) + j - 1;
        const uint32_t xo = cc >= 0 && idx8 >= 0 && cc * CH + j * K < L
                                ? (o_xl + (uint32_t)((idx8 >> 8) * (int)p.x_chunk_stride + 2 * N + (idx8 & 255) * 4)) * 4u
                                : 0x80000000u;
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xo, n0 * 1024, 0));
    };
    auto request_row = [&](int cc) __attribute__((always_inline)) {
        const int ll = cc * CH + j * K;
        const bool v = cc >= 0 && ll < L && row_ok;
        const uint32_t pl = REV ? L - ll - K : ll;
        ld8(pu, rs_u, voff(p.u_d_stride, pl, v), so_u);
        ld8(pdt, rs_dt, voff(p.delta_d_stride, pl, v), so_dt);
        ld8(pdo, rs_do, voff(q.dout_d_stride, pl, v), so_do);
        ld8(pz, rs_z, voff(p.z_d_stride, pl, v), so_z);
        if (DZM == 1) {
            ld8(pout, rs_out, voff(p.out_d_stride, pl, v), so_out);
            ld8(pout2, rs_out2, voff(out2_d_stride, pl, v), so_out2);
        }
    };

    request_row(n_c - 1);
    f32x4 xnext = request_x(n_c - 1, 0);
    stage_issue(n_c - 1);
    stage_commit();
    {   // records: {A log2(e), -, a entering from the right, adjoint entering from the right} per (row, state)
        const float A_mine = static_cast<const float*>(p.A)[(int64_t)dc * p.A_d_stride + (int64_t)j * p.A_dstate_stride];
        rec4[j] = f32x4{A_mine * kLog2e, 0.f, 1.f, 0.f};
    }
    lds_barrier_b();
    f32x4 bc = rec4[0];
    const bool is_first = j == 0, is_last = j == 15;
    for (int c = n_c - 1; c >= 0; --c) {
        asm volatile("" : "+v"(rr));
        const int l0 = c * CH + j * K;
        const bool ok = l0 < L && row_ok;
        const uint32_t pl0 = REV ? L - l0 - K : l0;
        const int rd_lo = c * CH + rd_pos;
        const uint32_t rd_vo_c = rd_lo < L ? (uint32_t)(REV ? L - 1 - rd_lo : rd_lo) * 4u : 0x80000000u;
        f2 dl2[K / 2], dlu2[K / 2], dy2[K / 2];
        float sdt = 0.f;             // sum of the lane's delta without the first one
        {
            const float bias = (*rowc).y;
            float dD_acc = 0.f;
            float dy[K], dlv[K];
            // (ONE branch on the launch-uniform flag around the whole loop: inside it the compiler kept a branch per element, and the
            // eight exp -> rcp -> log chains of a lane ran one behind the other)
            if (p.delta_softplus) {
#pragma unroll
                for (int i = 0; i < K; ++i) dlv[i] = softplusf_(pdt.at(i) + bias);
            } else {
#pragma unroll
                for (int i = 0; i < K; ++i) dlv[i] = pdt.at(i) + bias;
            }
#pragma unroll
            for (int i = 0; i < K; ++i) {
                dy[i] = ok ? pdo.at(i) : 0.f;   // past the end: c = 0, a = 1 (identity for the suffix scan)
                dlv[i] = ok ? dlv[i] : 0.f;
                if (i > 0) sdt += dlv[i];
            }
            {
                float dzv[K];
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    const float zv = pz.at(i);
                    const float s = sigmoidf_(zv);
                    if (DZM == 1) dzv[i] = dy[i] * (pout.at(i) + pout2.at(i)) * s * (1.f + zv * (1.f - s));
                    dy[i] *= zv * s;
                }
                if (DZM == 1) st8(dzv, rs_dz, voff(q.dz_d_stride, pl0, ok), so_dz, std::integral_constant<int, 0>{});
            }
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const float uv = pu.at(i);
                dl2[i / 2][i % 2] = dlv[i];
                dy2[i / 2][i % 2] = dy[i];
                dlu2[i / 2][i % 2] = dlv[i] * uv;
                dD_acc = fmaf(dy[i], uv, dD_acc);
            }
            const float dD_row = row_allsum_b(dD_acc);
            if (j == 0) ((lds_f32*)rowc)[2] += dD_row;
        }
        VMS_PT(5);   // prologue
        f2 S1[K / 2], S2[K / 2];  // per element: sum_n g B  /  sum_n A log2(e) g a x_{i-1}
#pragma unroll
        for (int k = 0; k < K / 2; ++k) {
            S1[k] = f2{0.f, 0.f};
            S2[k] = f2{0.f, 0.f};
        }
#define VMS_EL(arr, i) arr[(i) / 2][(i) % 2]
        auto do_state = [&](const int n, const float xseed, auto land_tag, auto last_tag) __attribute__((always_inline)) {
            const int buf = n & 1;
            // fp32 B / C of this state, shared by the workgroup's rows
            const lds_f4* bsrc = bc4 + (n * CH) / 4 + j;
            const f32x4 b0 = bsrc[0], b1 = bsrc[16], c0 = bsrc[N * CH / 4], c1 = bsrc[N * CH / 4 + 16];
            const float An = bc.x, anx_n = bc.z, gin = bc.w;
            bc = rec4[(n + 1) & (N - 1)];
            const f2 An2 = f2{An, An};
            f2 Bn2[K / 2], g2[K / 2], a2[K / 2];
#pragma unroll
            for (int k = 0; k < K / 2; ++k) {
                Bn2[k] = k == 0 ? f2{b0.x, b0.y} : k == 1 ? f2{b0.z, b0.w} : k == 2 ? f2{b1.x, b1.y} : f2{b1.z, b1.w};
                const f2 t = dl2[k] * An2;
                a2[k] = f2{fast_exp2(t.x), fast_exp2(t.y)};
                g2[k] = (k == 0 ? f2{c0.x, c0.y} : k == 1 ? f2{c0.z, c0.w} : k == 2 ? f2{c1.x, c1.y} : f2{c1.z, c1.w}) * dy2[k];
            }
            // adjoint: the lane's own suffix (from 0), the row scan of the lane aggregates, then the recurrence proper
            const float a_right = bdpp<DPP_ROW_SHL1>(anx_n, a2[0].x);  // lane 15 of the row <- next chunk
            float rg = 0.f;
#pragma unroll
            for (int i = K - 1; i >= 0; --i) rg = fmaf(i == K - 1 ? a_right : VMS_EL(a2, i + 1), rg, VMS_EL(g2, i));
            float ra = fast_exp2(sdt * An) * a_right;
            rg = fmaf(ra, is_last ? gin : 0.f, rg);
            row_scan_suffix_b(ra, rg);
            float grun = bdpp<DPP_ROW_SHL1>(gin, rg);
            if (is_first) *(lds_f2*)(rec1 + 4 * n + 2) = f2{a2[0].x, rg};
#pragma unroll
            for (int i = K - 1; i >= 0; --i) {
                grun = fmaf(i == K - 1 ? a_right : VMS_EL(a2, i + 1), grun, VMS_EL(g2, i));
                VMS_EL(g2, i) = grun;
            }
            VMS_PT(0);   // B / C, exponentials, the adjoint chains and the row scan
            // the state recurrence, element pair by element pair, with everything that consumes it
            f2 dA2 = f2{0.f, 0.f};
            float xrun = xseed;
#pragma unroll
            for (int k = 0; k < K / 2; ++k) {
                const f2 bb = dlu2[k] * Bn2[k];
                const float ax0 = a2[k].x * xrun;
                const float x0 = ax0 + bb.x;
                const float ax1 = a2[k].y * x0;
                const float x1 = ax1 + bb.y;
                xrun = x1;
                const f2 gg = g2[k];
                const f2 gax = gg * f2{ax0, ax1};   // g a_i x_{i-1}
                S1[k] = pk_fma_b(gg, Bn2[k], S1[k]);
                S2[k] = pk_fma_b(An2, gax, S2[k]);
                dA2 = pk_fma_b(dl2[k], gax, dA2);
                const f2 dBv = gg * dlu2[k], dCv = dy2[k] * f2{x0, x1};
                g2[k] = dBv;    // the products take the places of g and a
                a2[k] = dCv;
            }
            // the 4 rows of the wave summed through LDS: piece (tensor, element half) of every lane, then DPP row rho reads
            // piece rho of the four lanes that share its position group: tensor rho >> 1, elements 4 (rho & 1) + k, all 4 rows
            tr4[lane] = __builtin_shufflevector(g2[0], g2[1], 0, 1, 2, 3);
            tr4[kWave + lane] = __builtin_shufflevector(g2[2], g2[3], 0, 1, 2, 3);
            tr4[2 * kWave + lane] = __builtin_shufflevector(a2[0], a2[1], 0, 1, 2, 3);
            tr4[3 * kWave + lane] = __builtin_shufflevector(a2[2], a2[3], 0, 1, 2, 3);
            // the next chunk's row data goes out HERE in the chunk's last state, into the registers the recurrence just vacated: it
            // travels during the sums below, the barrier and the epilogue (requested behind the loop, every chunk waited for HBM)
            if constexpr (decltype(last_tag)::value) request_row(c - 1);
            const float dA_tot = row_allsum_b(dA2.x + dA2.y);
            if (j == n) dAacc += dA_tot;
            VMS_PT(1);   // the state recurrence + products + LDS writes + dA
            // One vmcnt covers loads, stores and atomics, in issue order: the checkpoint request that went out ahead of this state
            // lands HERE, in front of this state's atomic -- left to the compiler its wait sat behind the atomic (vmcnt(0) at the
            // loop's back edge), and every state group waited for an L2 atomic's round trip, the whole workgroup with it.
            if constexpr (decltype(land_tag)::value) asm volatile("" ::"v"(xnext));
            // workgroup-level sum of the PREVIOUS state's partials (other slab buffer), for the threads whose turn it is
            const bool mine = W == 4 || ((n - 1) & 1) == rd_turn;   // wave-uniform
            if (mine) {
                const lds_f32* src = rd_src + (buf ^ 1) * kSlab;
                float pv[W];
#pragma unroll
                for (int w = 0; w < W; ++w) pv[w] = src[w * (4 * kWave)];
                float tsum = pv[0] + pv[1];
#pragma unroll
                for (int w = 2; w < W; w += 2) tsum += pv[w] + pv[w + 1];
                if (rd_tenu) __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(tsum, rs_dC, rd_vo, rd_so, 0);
                else __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(tsum, rs_dB, rd_vo, rd_so, 0);
            }
            VMS_PT(2);   // workgroup-level sum of the previous state + atomic
            {
                const lds_f4* const trs = tr4 + r * kWave + j;
                const f32x4 q0 = trs[0], q1 = trs[16], q2 = trs[32], q3 = trs[48];
                const f2 o0 = (f2{q0.x, q0.y} + f2{q2.x, q2.y}) + (f2{q1.x, q1.y} + f2{q3.x, q3.y});
                const f2 o1 = (f2{q0.z, q0.w} + f2{q2.z, q2.w}) + (f2{q1.z, q1.w} + f2{q3.z, q3.w});
                slab4[(buf * kSlab + quad * (4 * kWave)) / 4 + lane] = __builtin_shufflevector(o0, o1, 0, 1, 2, 3);
            }
            rd_vo = rd_vo_c;
            rd_so = (uint32_t)n * rd_sstride;
            VMS_PT(3);   // the 4 rows through LDS, slab write
            lds_barrier_b();  // this state's partials written by all waves; the other buffer free again
            VMS_PT(4);   // barrier
        };
        asm volatile("" ::"v"(xnext));   // (landed in the previous chunk's last state; the first chunk's: here, not at the loop header)
#pragma unroll 1
        for (int n0 = 0; n0 < N; n0 += 4) {
            const f32x4 xc = xnext;
            // the next chunk's B / C pieces (PPT 16-byte registers per thread) go out four states ahead of their conversion: behind
            // the loop the staging commit -- and through the barrier behind it the whole workgroup -- waited for HBM every chunk
            if (n0 == N - 4) stage_issue(c - 1);
            do_state(n0, xc.x, std::false_type{}, std::false_type{});
            do_state(n0 + 1, xc.y, std::false_type{}, std::false_type{});
            do_state(n0 + 2, xc.z, std::false_type{}, std::false_type{});
            xnext = n0 + 4 < N ? request_x(c, n0 + 4) : request_x(c - 1, 0);
            do_state(n0 + 3, xc.w, std::true_type{}, std::false_type{});
        }
#undef VMS_EL
        VMS_PT(6);   // (loop overhead)
        request_row(c - 1);
        // the next chunk's row data and B / C pieces travel during the epilogue, the staging commit and the barrier

        {
            const float Dd = (*rowc).x;
            float dbias_acc = 0.f;
            float duv[K], ddl[K], sgv[K];
            if (p.delta_softplus) {   // (one branch for the eight elements, as in the prologue)
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    const float dl = dl2[i / 2][i % 2];
                    sgv[i] = dl < 9.765625e-4f ? dl * fmaf(-0.5f, dl, 1.f) : 1.f - fast_exp(-dl);
                }
            } else {
#pragma unroll
                for (int i = 0; i < K; ++i) sgv[i] = 1.f;
            }
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const float dl = dl2[i / 2][i % 2];
                const float s1 = S1[i / 2][i % 2], s2 = S2[i / 2][i % 2] * 0.6931471805599453f;
                // Neither u nor the softplus derivative crossed the state loop in registers (8 of the 128): both come back from
                // delta.  sigmoid(t) = 1 - exp(-softplus(t)) (exact identity; 1 above the reference's threshold of 20 by itself), as
                // delta (1 - delta / 2) where the subtraction would cancel; u = (delta u) / delta, 0 where delta underflowed to 0
                // (there sigmoid is 0 too: the reference's ddelta is ~1e-45 x (...) = 0).
                const float sg = sgv[i];
                const float uv = dl > 0.f ? dlu2[i / 2][i % 2] * fast_rcp(dl) : 0.f;
                duv[i] = fmaf(dl, s1, Dd * dy2[i / 2][i % 2]);
                ddl[i] = fmaf(uv, s1, s2) * sg;
                dbias_acc += ok ? ddl[i] : 0.f;
            }
            st8(duv, rs_du, voff(q.du_d_stride, pl0, ok), so_du, std::integral_constant<int, 2>{});
            st8(ddl, rs_ddt, voff(q.ddelta_d_stride, pl0, ok), so_ddt, std::integral_constant<int, 2>{});
            const float db_row = row_allsum_b(dbias_acc);
            if (j == 0) ((lds_f32*)rowc)[3] += db_row;
        }
        stage_commit();   // every wave is past its last B / C read of this chunk (the barrier of the last state)
        lds_barrier_b();
        VMS_PT(7);   // epilogue + staging commit + barrier
    }
    // the last state (buffer (N - 1) & 1) is still in the slab
    if (W == 4 || ((N - 1) & 1) == rd_turn) {
        const lds_f32* src = rd_src + ((N - 1) & 1) * kSlab;
        float tsum = src[0];
#pragma unroll
        for (int w = 1; w < W; ++w) tsum += src[w * (4 * kWave)];
        if (rd_tenu) __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(tsum, rs_dC, rd_vo, rd_so, 0);
        else __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(tsum, rs_dB, rd_vo, rd_so, 0);
    }
#ifdef VMS_O4_PROF
    if (prof && lane == 0) {
        const int wv = (int)(blockIdx.x * W + quad);
#pragma unroll
        for (int k = 0; k < 8; ++k) prof[wv * 8 + k] = pt[k];
    }
#endif
    if (row_ok) {
        const f32x4 rc = *rowc;
        if (q.dD && j == 0) atomicAdd(q.dD + d, rc.z);
        if (q.ddelta_bias && j == 0) atomicAdd(q.ddelta_bias + d, rc.w);
        atomicAdd(q.dA + (int64_t)d * q.dA_d_stride + (int64_t)j * q.dA_dstate_stride, dAacc);
    }
}

// the first half of the grid runs qa left-to-right and writes the whole dz, the second half runs qb right-to-left
template <typename T, int W>
__global__ __launch_bounds__(W* kWave, 4) void scan_bwd_o4_dual_kernel(const vms_scan_bwd_params qa, const vms_scan_bwd_params qb,
                                                                       unsigned long long* __restrict__ prof) {
    const int half = gridDim.x >> 1;
    if ((int)blockIdx.x < half) {
        scan_bwd_o4_body<T, false, W, 1>(qa, blockIdx.x, static_cast<const T*>(qb.f.out), qb.f.out_batch_stride, qb.f.out_d_stride, prof);
    } else {
        scan_bwd_o4_body<T, true, W, 2>(qb, blockIdx.x - half, nullptr, 0, 0, prof);
    }
}

template <typename T>
static int launch_o4_dual(const vms_scan_bwd_params& a, const vms_scan_bwd_params& b, hipStream_t stream) {
    const vms_scan_fwd_params& p = a.f;
    const int n8 = p.batch * ((p.dim + 31) / 32), n16 = p.batch * ((p.dim + 15) / 16), cus = device_cu_count();
    // 4-wave workgroups (44 KB of LDS: three per CU) while the whole grid is resident at once; beyond that 8-wave workgroups
    // (two per CU = 4 waves per SIMD: the benchmark shape's 512 workgroups are exactly one such round)
    const bool w8 = 2 * n16 > 3 * cus;
    static PerDeviceOnce attr_once;
    const hipError_t arc = attr_once.run([&]() -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&scan_bwd_o4_dual_kernel<T, 8>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)O4<8>::kSmem);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&scan_bwd_o4_dual_kernel<T, 4>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)O4<4>::kSmem);
        return e;
    });
    if (arc != hipSuccess) {
        set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed: %s", hipGetErrorString(arc));
        return VMS_ERR_LAUNCH;
    }
    unsigned long long* prof = nullptr;
#ifdef VMS_O4_PROF   /* diagnostic builds only: the C ABI proper reads no environment */
    if (const char* e = getenv("VMS_O4_PROF_PTR")) prof = reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 0));
#endif
    if (w8) {
        const dim3 grid(2 * n8), block(8 * kWave);
        hipLaunchKernelGGL((scan_bwd_o4_dual_kernel<T, 8>), grid, block, O4<8>::kSmem, stream, a, b, prof);
    } else {
        const dim3 grid(2 * p.batch * ((p.dim + 15) / 16)), block(4 * kWave);
        hipLaunchKernelGGL((scan_bwd_o4_dual_kernel<T, 4>), grid, block, O4<4>::kSmem, stream, a, b, prof);
    }
    VMS_LAUNCH_CHECK();
    set_last_kernel(w8 ? "scan_bwd_o4_dual_w8" : "scan_bwd_o4_dual_w4");
    return VMS_OK;
}

// called by launch_scan_bwd_pair_dual (selective_scan_bwd_pair.hip) for problems scan_bwd_pair_dual_fusable admits
bool scan_bwd_o4_dual_takes(const vms_scan_bwd_params& a, const vms_scan_bwd_params& b) {
    auto xl = [](const vms_scan_fwd_params& p) {
        return p.x_has_sub == 3 && (int64_t)p.dim * p.n_chunks * p.x_chunk_stride * 4 < ((int64_t)1 << 31);
    };
    return (a.f.dtype == VMS_BF16 || a.f.dtype == VMS_F16) && xl(a.f) && xl(b.f) && a.f.impl == VMS_IMPL_OCC4 && b.f.impl == VMS_IMPL_OCC4;
}

int launch_scan_bwd_o4_dual(const vms_scan_bwd_params& a, const vms_scan_bwd_params& b, hipStream_t stream) {
    switch (a.f.dtype) {
        case VMS_BF16: return launch_o4_dual<bf16_t>(a, b, stream);
        case VMS_F16: return launch_o4_dual<f16_t>(a, b, stream);
        default: set_error("dual backward: 16-bit activations only"); return VMS_ERR_UNSUPPORTED;
    }
}

}  // namespace vms
