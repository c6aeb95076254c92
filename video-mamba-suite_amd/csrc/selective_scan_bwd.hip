// selective_scan_bwd.hip -- backward selective scan for gfx950 (wave64).
//
// Replaces selective_scan_bwd_kernel (mamba/csrc/selective_scan/selective_scan_bwd_kernel.cuh:
// 75-489), BlockReverseScan (reverse_scan.cuh:202-401) and the launcher (:491-531).
// Design (DESIGN.md "scan backward"):
//   * same decomposition as the forward: one wave = one (batch, dim) row, 64*K-element chunks
//     walked from the END of the sequence to the start, lane j owns K consecutive elements.
//   * per state n the forward recurrence is re-scanned from the 1024-element checkpoints the
//     forward wrote into x (vms_hip.h), then the adjoint recurrence
//         g_l = C_l dy_l + a_{l+1} g_{l+1}
//     is scanned right-to-left with a DPP suffix scan; both keep their K values in registers.
//   * dB / dC (sums over the dims of a group) are first reduced over the 4 rows of the
//     workgroup in a padded LDS tile (ds_add_f32), then flushed with coalesced fp32 global
//     atomics -- 4x fewer L2 atomics than one per (row, state, position) as in the reference
//     (selective_scan_bwd_kernel.cuh:297-316).
//   * dA, dD, ddelta_bias, constant-B/C gradients: wave reduction, one atomic per row.
#include <stdlib.h>

#include "vms_common.h"

namespace vms {

int validate_scan_common(const vms_scan_fwd_params& p);
bool scan_fwd_vec_ok(const vms_scan_fwd_params& p);
bool scan_bwd_pair_eligible(const vms_scan_bwd_params& q, bool vec);
int launch_scan_bwd_pair(const vms_scan_bwd_params& q, hipStream_t stream);
bool scan_bwd_pair_dual_fusable(const vms_scan_bwd_params& a, const vms_scan_bwd_params& b);
int launch_scan_bwd_pair_dual(const vms_scan_bwd_params& a, const vms_scan_bwd_params& b, hipStream_t stream);
int scan_bwd_pair_segments(const vms_scan_bwd_params& q);
int64_t scan_bwd_pair_ws_bytes(const vms_scan_bwd_params& q);

int launch_scan_bwd_complex(const vms_scan_bwd_params& q, bool vec, hipStream_t stream);   // selective_scan_complex.hip
bool scan_bwd_short_eligible(const vms_scan_bwd_params& q);                                  // selective_scan_short.hip
bool scan_short_takes(const vms_scan_fwd_params& p);
int launch_scan_bwd_short(const vms_scan_bwd_params& q, hipStream_t stream);
int64_t scan_bwd_short_ws_bytes(const vms_scan_bwd_params& q);
int scan_short_nseg(const vms_scan_fwd_params& p);
bool scan_bwd_short_has_ws(const vms_scan_bwd_params& q);

constexpr int kBwdRows = 4;
constexpr int kTilePad = 65;  // tile index = i * 65 + lane : conflict-free ds_add, 2-way flush

template <typename T, int K, bool VB, bool VC, bool HZ, bool VEC>
__global__ __launch_bounds__(kBwdRows* kWave) void scan_bwd_kernel(const vms_scan_bwd_params q, const int dbg) {
    const vms_scan_fwd_params& p = q.f;
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tiles = (p.dim + kBwdRows - 1) / kBwdRows;
    const int b = blockIdx.x / tiles;
    const int d0 = (blockIdx.x - b * tiles) * kBwdRows;
    const int dpg = p.dim / p.n_groups;
    const bool row_ok = d0 + wave < p.dim;
    const int d = row_ok ? d0 + wave : p.dim - 1;
    const int g = d / dpg;
    // all rows of this workgroup in one group -> dB/dC can be pre-reduced in LDS
    const int d_last = min(d0 + kBwdRows, p.dim) - 1;
    const bool same_group = (d0 / dpg) == (d_last / dpg);
    const int g0 = d0 / dpg;
    const int L = p.seqlen, N = p.dstate;
    const bool rev = p.reverse != 0;
    constexpr int CS = kWave * K;
    constexpr int TILE = K * kTilePad;

    // LDS carve: [2 buffers][dB tile | dC tile] shared by the WG, then per-wave state arrays
    lds_f32* tiles_lds = (lds_f32*)smem;                                     // 2 * 2 * TILE floats
    volatile lds_f32* wv = (lds_f32*)smem + 4 * TILE + wave * (5 * N);       // per wave state arrays
    volatile lds_f32* gcarry = wv;          // adjoint entering from the right, per state
    volatile lds_f32* anext = wv + N;       // a of the first element of the chunk to the right
    volatile lds_f32* dA_acc = wv + 2 * N;  // dA accumulated over chunks
    volatile lds_f32* dBc_acc = wv + 3 * N; // gradient of a constant (dim, dstate) B
    volatile lds_f32* dCc_acc = wv + 4 * N; // gradient of a constant (dim, dstate) C
    for (int n = lane; n < N; n += kWave) {
        gcarry[n] = 0.f;
        anext[n] = 1.f;
        dA_acc[n] = 0.f;
        dBc_acc[n] = 0.f;
        dCc_acc[n] = 0.f;
    }
    for (int i = threadIdx.x; i < 4 * TILE; i += blockDim.x) tiles_lds[i] = 0.f;
    __syncthreads();

    const T* u = static_cast<const T*>(p.u) + (int64_t)b * p.u_batch_stride + (int64_t)d * p.u_d_stride;
    const T* dt = static_cast<const T*>(p.delta) + (int64_t)b * p.delta_batch_stride + (int64_t)d * p.delta_d_stride;
    const T* dout = static_cast<const T*>(q.dout) + (int64_t)b * q.dout_batch_stride + (int64_t)d * q.dout_d_stride;
    T* du = static_cast<T*>(q.du) + (int64_t)b * q.du_batch_stride + (int64_t)d * q.du_d_stride;
    T* ddelta = static_cast<T*>(q.ddelta) + (int64_t)b * q.ddelta_batch_stride + (int64_t)d * q.ddelta_d_stride;
    const T* z = HZ ? static_cast<const T*>(p.z) + (int64_t)b * p.z_batch_stride + (int64_t)d * p.z_d_stride : nullptr;
    const T* outp = HZ ? static_cast<const T*>(p.out) + (int64_t)b * p.out_batch_stride + (int64_t)d * p.out_d_stride : nullptr;
    T* dz = HZ ? static_cast<T*>(q.dz) + (int64_t)b * q.dz_batch_stride + (int64_t)d * q.dz_d_stride : nullptr;
    T* out_z = (HZ && p.out_z) ? static_cast<T*>(p.out_z) + (int64_t)b * p.out_z_batch_stride + (int64_t)d * p.out_z_d_stride : nullptr;
    const float* A = static_cast<const float*>(p.A) + (int64_t)d * p.A_d_stride;
    const T* Bv = VB ? static_cast<const T*>(p.B) + (int64_t)b * p.B_batch_stride + (int64_t)g * p.B_group_stride : nullptr;
    const T* Cv = VC ? static_cast<const T*>(p.C) + (int64_t)b * p.C_batch_stride + (int64_t)g * p.C_group_stride : nullptr;
    const float* Bc = !VB ? static_cast<const float*>(p.B) + (int64_t)d * p.B_d_stride : nullptr;
    const float* Cc = !VC ? static_cast<const float*>(p.C) + (int64_t)d * p.C_d_stride : nullptr;
    const int64_t xpitch = p.x_chunk_stride ? p.x_chunk_stride : 2 * N;
    const float* xck = p.x ? static_cast<const float*>(p.x) + ((int64_t)b * p.dim + d) * p.n_chunks * xpitch : nullptr;
    const float Dd = p.D ? static_cast<const float*>(p.D)[d] : 0.f;
    const float bias = p.delta_bias ? static_cast<const float*>(p.delta_bias)[d] : 0.f;
    float* dBg = VB ? q.dB + (int64_t)b * q.dB_batch_stride : nullptr;  // + group, state, l below
    float* dCg = VC ? q.dC + (int64_t)b * q.dC_batch_stride : nullptr;

    float dD_acc = 0.f, dbias_acc = 0.f;
    int buf = 0;  // LDS tile double buffer, flips once per (chunk, state)

    const int n_kchunks = (L + CS - 1) / CS;
    for (int c = n_kchunks - 1; c >= 0; --c) {
        const int l0 = c * CS + lane * K;
        const int nv = row_ok ? L - l0 : 0;
        const int Lr = row_ok ? L : 0;  // rows past dim read nothing
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
            float zv[K], ov[K];
            load_dir<T, K, VEC>(z, l0, Lr, rev, zv);
            load_dir<T, K, VEC>(outp, l0, Lr, rev, ov);
            float dzv[K];
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const float s = sigmoidf_(zv[i]);
                const float silu = zv[i] * s;
                dzv[i] = dy[i] * ov[i] * s * (1.f + zv[i] * (1.f - s));
                dy[i] *= silu;
                ov[i] *= silu;
            }
            if (q.dz_accumulate) {  // dz += (vms_hip.h)
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
        // checkpoint holding the state at the start of this chunk (position c*CS)
        const int pos0 = c * CS;
        const float* xin = nullptr;
        int xoff = 0;
        if (pos0 > 0) {
            const int blk = pos0 / 2048;
            if (pos0 % 2048 == 0) { xin = xck + (int64_t)(blk - 1) * xpitch; xoff = 1; }
            else { xin = xck + (int64_t)blk * xpitch; xoff = 0; }  // pos0 % 2048 == 1024
        }
        for (int n = 0; n < N; ++n) {
            const float Araw = A[n * p.A_dstate_stride];
            const float An = Araw * kLog2e;
            float Bn[K], Cn[K];
            if (VB) load_dir<T, K, VEC>(Bv + (int64_t)n * p.B_dstate_stride, l0, Lr, rev, Bn);
            if (VC) load_dir<T, K, VEC>(Cv + (int64_t)n * p.C_dstate_stride, l0, Lr, rev, Cn);
            const float bconst = VB ? 1.f : Bc[n * p.B_dstate_stride];
            const float cconst = VC ? 1.f : Cc[n * p.C_dstate_stride];
            // ---- forward re-scan: x_i for the lane's K elements ----
            float a[K], xs[K];
            float pa = 1.f, px = 0.f;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                a[i] = fast_exp2(dl[i] * An);
                const float bxi = dl[i] * uv[i] * (VB ? Bn[i] : bconst);
                px = fmaf(a[i], px, bxi);
                pa *= a[i];
            }
            wave_scan_inclusive(pa, px);
            const float ea = dpp_mov<DPP_WAVE_SHR1, 0xf>(1.f, pa);
            const float ex = dpp_mov<DPP_WAVE_SHR1, 0xf>(0.f, px);
            const float hin = xin ? xin[2 * n + xoff] : 0.f;
            float xrun = fmaf(ea, hin, ex);
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const float bxi = dl[i] * uv[i] * (VB ? Bn[i] : bconst);
                xrun = fmaf(a[i], xrun, bxi);
                xs[i] = xrun;
            }
            // ---- adjoint scan, right to left: g_i = C_i dy_i + alpha_i g_{i+1}, alpha_i = a_{i+1}
            const float a_right_lane = dpp_mov<DPP_WAVE_SHL1, 0xf>(anext[n], a[0]);  // lane 63 <- next chunk
            float gl[K];
            float ra = 1.f, rg = 0.f;
#pragma unroll
            for (int i = K - 1; i >= 0; --i) {
                const float alpha = i == K - 1 ? a_right_lane : a[i + 1];
                const float ci = dy[i] * (VC ? Cn[i] : cconst);
                rg = fmaf(alpha, rg, ci);
                ra *= alpha;
            }
            wave_scan_inclusive_reverse(ra, rg);
            const float esa = dpp_mov<DPP_WAVE_SHL1, 0xf>(1.f, ra);
            const float esx = dpp_mov<DPP_WAVE_SHL1, 0xf>(0.f, rg);
            const float gin = gcarry[n];
            float grun = fmaf(esa, gin, esx);
            const float gout = fmaf(readlane_f(ra, 0), gin, readlane_f(rg, 0));
            const float a_first = readlane_f(a[0], 0);
            if (lane == 0) {
                gcarry[n] = gout;
                anext[n] = a_first;
            }
            float dA_loc = 0.f, dBc_loc = 0.f, dCc_loc = 0.f;
            lds_f32* tb = tiles_lds + buf * 2 * TILE;
            lds_f32* tc = tb + TILE;
#pragma unroll
            for (int i = K - 1; i >= 0; --i) {
                const float alpha = i == K - 1 ? a_right_lane : a[i + 1];
                const float ci = dy[i] * (VC ? Cn[i] : cconst);
                grun = fmaf(alpha, grun, ci);
                gl[i] = grun;
            }
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const float gx = gl[i];
                const float du_i = dl[i] * uv[i];
                const float Bi = VB ? Bn[i] : bconst;
                const float ddelta_u = gx * Bi;
                const float ax = xs[i] - du_i * Bi;  // a_i * x_{i-1}
                duv[i] = fmaf(ddelta_u, dl[i], duv[i]);
                ddl[i] = fmaf(ddelta_u, uv[i], ddl[i]);
                ddl[i] = fmaf(gx * Araw, ax, ddl[i]);
                dA_loc = fmaf(gx * dl[i], ax, dA_loc);
                const float dBi = gx * du_i;
                const float dCi = dy[i] * xs[i];
                if (VB) {
                    if (same_group) { if (i < nv && !(dbg & 2)) lds_atomic_add(&tb[i * kTilePad + lane], dBi); }
                    else if (i < nv) atomicAdd(dBg + (int64_t)g * q.dB_group_stride + (int64_t)n * q.dB_dstate_stride + (rev ? L - 1 - (l0 + i) : l0 + i), dBi);
                } else {
                    dBc_loc += dBi;
                }
                if (VC) {
                    if (same_group) { if (i < nv && !(dbg & 2)) lds_atomic_add(&tc[i * kTilePad + lane], dCi); }
                    else if (i < nv) atomicAdd(dCg + (int64_t)g * q.dC_group_stride + (int64_t)n * q.dC_dstate_stride + (rev ? L - 1 - (l0 + i) : l0 + i), dCi);
                } else {
                    dCc_loc += dCi;
                }
            }
            const float dA_tot = wave_sum(dA_loc);
            if (lane == 0) dA_acc[n] += dA_tot;
            if (!VB) {
                const float t = wave_sum(dBc_loc);
                if (lane == 0) dBc_acc[n] += t;
            }
            if (!VC) {
                const float t = wave_sum(dCc_loc);
                if (lane == 0) dCc_acc[n] += t;
            }
            if ((VB || VC) && same_group) {
                if (!(dbg & 4)) __syncthreads();  // every row's contribution to (chunk, state) is in tile `buf`
                const int lim = (dbg & 1) ? 0 : min(CS, L - c * CS);
                for (int j = threadIdx.x; j < lim; j += blockDim.x) {
                    const int ti = (j % K) * kTilePad + (j / K);
                    if (VB) {
                        atomicAdd(dBg + (int64_t)g0 * q.dB_group_stride + (int64_t)n * q.dB_dstate_stride + (rev ? L - 1 - (c * CS + j) : c * CS + j), tb[ti]);
                        tb[ti] = 0.f;
                    }
                    if (VC) {
                        atomicAdd(dCg + (int64_t)g0 * q.dC_group_stride + (int64_t)n * q.dC_dstate_stride + (rev ? L - 1 - (c * CS + j) : c * CS + j), tc[ti]);
                        tc[ti] = 0.f;
                    }
                }
                buf ^= 1;  // the other buffer was flushed+zeroed one barrier ago
            }
        }
        // softplus chain (bwd_kernel.cuh:439-452) and stores
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
    if (row_ok) {
        if (q.dD) {
            const float t = wave_sum(dD_acc);
            if (lane == 0) atomicAdd(q.dD + d, t);
        }
        if (q.ddelta_bias) {
            const float t = wave_sum(dbias_acc);
            if (lane == 0) atomicAdd(q.ddelta_bias + d, t);
        }
        for (int n = lane; n < N; n += kWave) {
            atomicAdd(q.dA + (int64_t)d * q.dA_d_stride + (int64_t)n * q.dA_dstate_stride, dA_acc[n]);
            if (!VB) atomicAdd(q.dB + (int64_t)d * q.dB_d_stride + (int64_t)n * q.dB_dstate_stride, dBc_acc[n]);
            if (!VC) atomicAdd(q.dC + (int64_t)d * q.dC_d_stride + (int64_t)n * q.dC_dstate_stride, dCc_acc[n]);
        }
    }
}

template <typename T, int K, bool VB, bool VC, bool HZ>
static int launch_bwd(const vms_scan_bwd_params& q, bool vec, hipStream_t stream) {
    const vms_scan_fwd_params& p = q.f;
    const int tiles = (p.dim + kBwdRows - 1) / kBwdRows;
    dim3 grid(p.batch * tiles), block(kBwdRows * kWave);
    const size_t smem = sizeof(float) * (4 * K * kTilePad + kBwdRows * 5 * p.dstate);
    const int dbg = 0;  // ablation bits of the kernel (profiling builds only)
    if (vec)
        hipLaunchKernelGGL((scan_bwd_kernel<T, K, VB, VC, HZ, true>), grid, block, smem, stream, q, dbg);
    else
        hipLaunchKernelGGL((scan_bwd_kernel<T, K, VB, VC, HZ, false>), grid, block, smem, stream, q, dbg);
    VMS_LAUNCH_CHECK();
    return VMS_OK;
}

template <typename T, int K>
static int dispatch_bwd(const vms_scan_bwd_params& q, bool vec, hipStream_t s) {
    const bool vb = q.f.is_variable_B, vc = q.f.is_variable_C, hz = q.f.z != nullptr;
#define VMS_CASE(B_, C_, Z_) \
    if (vb == B_ && vc == C_ && hz == Z_) return launch_bwd<T, K, B_, C_, Z_>(q, vec, s);
    VMS_CASE(true, true, true)
    VMS_CASE(true, true, false)
    VMS_CASE(true, false, true)
    VMS_CASE(true, false, false)
    VMS_CASE(false, true, true)
    VMS_CASE(false, true, false)
    VMS_CASE(false, false, true)
    VMS_CASE(false, false, false)
#undef VMS_CASE
    return VMS_ERR_INVALID_ARG;
}

}  // namespace vms

using namespace vms;

// scratch the paired kernel wants when it splits the sequence into ranges (few rows, long sequences): the (P, q)
// adjoint carries of selective_scan_bwd_pair.hip; 0 = no scratch needed for this problem
static inline int64_t round256b(int64_t n) { return (n + 255) & ~(int64_t)255; }
// reverse_from (ABI v5): the two sub-batches as two problems (dA, dD, ddelta_bias are shared accumulators)
static void scan_bwd_sub_batches(const vms_scan_bwd_params& q, vms_scan_bwd_params& lo, vms_scan_bwd_params& hi) {
    const vms_scan_fwd_params& p = q.f;
    const int rf = p.reverse_from, es = p.dtype == VMS_F32 ? 4 : 2;
    lo = q; hi = q;
    scan_fwd_sub_batches(p, lo.f, hi.f);
    auto offc = [](const void* ptr, int64_t bytes) -> const void* { return ptr ? static_cast<const char*>(ptr) + bytes : nullptr; };
    auto offm = [](void* ptr, int64_t bytes) -> void* { return ptr ? static_cast<char*>(ptr) + bytes : nullptr; };
    hi.dout = offc(q.dout, rf * q.dout_batch_stride * es);
    hi.du = offm(q.du, rf * q.du_batch_stride * es);
    hi.ddelta = offm(q.ddelta, rf * q.ddelta_batch_stride * es);
    hi.dz = offm(q.dz, rf * q.dz_batch_stride * es);
    if (p.is_variable_B) hi.dB = q.dB + rf * q.dB_batch_stride;
    if (p.is_variable_C) hi.dC = q.dC + rf * q.dC_batch_stride;
}

extern "C" int64_t vms_scan_bwd_workspace_bytes(const vms_scan_bwd_params* q) {
    if (q != nullptr && q->f.is_complex) return 0;
    if (q != nullptr && q->f.reverse_from > 0 && q->f.reverse_from < q->f.batch &&
        !(scan_impl_level(q->f) >= VMS_IMPL_PAIR && scan_bwd_pair_eligible(*q, true) && scan_bwd_pair_native_mixed(*q) && !scan_short_takes(q->f))) {
        vms_scan_bwd_params lo, hi;
        scan_bwd_sub_batches(*q, lo, hi);
        const int64_t wl = vms_scan_bwd_workspace_bytes(&lo), wh = vms_scan_bwd_workspace_bytes(&hi);
        return wl + wh > 0 ? round256b(wl) + wh : 0;
    }
    // the lane-per-row kernels: per-row dA / dD / ddelta_bias, summed by a second kernel instead of 1,568-way atomics
    if (q != nullptr && scan_impl_level(q->f) >= VMS_IMPL_PAIR && q->f.x_has_sub == 0 && scan_bwd_short_eligible(*q)) return scan_bwd_short_ws_bytes(*q);
    if (q == nullptr || scan_impl_level(q->f) < VMS_IMPL_PAIR || !scan_bwd_pair_eligible(*q, true)) return 0;
    return scan_bwd_pair_segments(*q) > 1 ? scan_bwd_pair_ws_bytes(*q) : 0;
}

extern "C" int vms_selective_scan_bwd(const vms_scan_bwd_params* qq, void* stream) {
    VMS_CHECK(qq != nullptr, "null params");
    const vms_scan_bwd_params& q = *qq;
    const vms_scan_fwd_params& p = q.f;
    if (int rc = validate_scan_common(p)) return rc;
    VMS_CHECK(q.dout && q.du && q.ddelta && q.dA && q.dB && q.dC, "dout, du, ddelta, dA, dB, dC are required");
    if (p.reverse_from != 0) {
        VMS_CHECK(p.reverse_from > 0 && p.reverse_from <= p.batch && p.reverse == 0, "reverse_from must be in (0, batch] with reverse == 0");
        const bool native = p.is_complex || (scan_impl_level(p) >= VMS_IMPL_PAIR && scan_bwd_pair_eligible(q, true) && scan_bwd_pair_native_mixed(q) &&
                                             !scan_short_takes(p));
        if (p.reverse_from < p.batch && !native) {
            vms_scan_bwd_params lo, hi;
            scan_bwd_sub_batches(q, lo, hi);
            const int64_t wl = round256b(vms_scan_bwd_workspace_bytes(&lo)), wh = vms_scan_bwd_workspace_bytes(&hi);
            if (p.workspace != nullptr && p.workspace_bytes >= wl + wh) {
                if (wl > 0) { lo.f.workspace = p.workspace; lo.f.workspace_bytes = wl; }
                if (wh > 0) { hi.f.workspace = static_cast<char*>(p.workspace) + wl; hi.f.workspace_bytes = wh; }
            }
            if (int rc = vms_selective_scan_bwd(&lo, stream)) return rc;
            return vms_selective_scan_bwd(&hi, stream);
        }
    }
    VMS_CHECK(p.is_complex || p.x != nullptr || p.seqlen <= 1024, "x (scan checkpoints) is required when seqlen > 1024");
    if (p.z) {
        VMS_CHECK(p.out != nullptr, "out is required when z is given");
        VMS_CHECK(q.dz != nullptr, "dz is required when z is given");
    }
    VMS_CHECK((p.D == nullptr) == (q.dD == nullptr), "dD must be given iff D is given");
    VMS_CHECK((p.delta_bias == nullptr) == (q.ddelta_bias == nullptr), "ddelta_bias must be given iff delta_bias is given");
    const int es = p.dtype == VMS_F32 ? 4 : 2;
    bool vec = scan_fwd_vec_ok(p) && aligned16(q.dout) && aligned16(q.du) && aligned16(q.ddelta) &&
               mult16(q.dout_batch_stride, es) && mult16(q.dout_d_stride, es) && mult16(q.du_batch_stride, es) &&
               mult16(q.du_d_stride, es) && mult16(q.ddelta_batch_stride, es) && mult16(q.ddelta_d_stride, es);
    if (p.z) vec = vec && aligned16(q.dz) && mult16(q.dz_batch_stride, es) && mult16(q.dz_d_stride, es);
    hipStream_t s = static_cast<hipStream_t>(stream);
    VMS_CHECK(scan_impl_valid(p.impl) && p.segments >= 0, "impl / segments out of range");
    if (p.is_complex) return launch_scan_bwd_complex(q, vec, s);
    const int level = scan_impl_level(p);
    // short rows (selective_scan_short.hip): the states are rebuilt from h = 0 in the lane, x is not read
    // (segmented rows, 17 .. 64 elements, need the workspace for the adjoint carries: without it the generic kernel below takes them)
    if (level >= VMS_IMPL_PAIR && p.x_has_sub == 0 && scan_bwd_short_eligible(q) && (scan_short_nseg(p) == 1 || scan_bwd_short_has_ws(q)))
        return launch_scan_bwd_short(q, s);
    if (level >= VMS_IMPL_PAIR && scan_bwd_pair_eligible(q, vec)) return launch_scan_bwd_pair(q, s);
    set_last_kernel("scan_bwd_generic");
    switch (p.dtype) {
        case VMS_F32: return dispatch_bwd<float, 16>(q, vec, s);
        case VMS_F16: return dispatch_bwd<f16_t, 16>(q, vec, s);
        default: return dispatch_bwd<bf16_t, 16>(q, vec, s);
    }
}

// Both directions of a bidirectional block (vms_hip.h): one grid when the pair qualifies, else the two single calls with the
// second adding its dz to the first's.
static bool scan_bwd_dual_checks_ok(const vms_scan_bwd_params& a, const vms_scan_bwd_params& b) {
    using namespace vms;
    const vms_scan_fwd_params &pa = a.f, &pb = b.f;
    if (pa.is_complex || pb.is_complex || validate_scan_common(pa) != VMS_OK || validate_scan_common(pb) != VMS_OK) return false;
    if (!(a.dout && a.du && a.ddelta && a.dA && a.dB && a.dC && b.dout && b.du && b.ddelta && b.dA && b.dB && b.dC)) return false;
    if ((pa.D == nullptr) != (a.dD == nullptr) || (pb.D == nullptr) != (b.dD == nullptr)) return false;
    if ((pa.delta_bias == nullptr) != (a.ddelta_bias == nullptr) || (pb.delta_bias == nullptr) != (b.ddelta_bias == nullptr)) return false;
    if (!scan_impl_valid(pa.impl) || pb.impl != pa.impl || pa.segments < 0 || pb.segments < 0) return false;
    return scan_impl_level(pa) >= VMS_IMPL_PAIR;
}

extern "C" int vms_scan_bwd_dual_fused(const vms_scan_bwd_params* a, const vms_scan_bwd_params* b) {
    if (a == nullptr || b == nullptr) return 0;
    return scan_bwd_dual_checks_ok(*a, *b) && vms::scan_bwd_pair_dual_fusable(*a, *b) ? 1 : 0;
}

extern "C" int vms_selective_scan_bwd_dual(const vms_scan_bwd_params* a, const vms_scan_bwd_params* b, void* stream) {
    using namespace vms;
    VMS_CHECK(a != nullptr && b != nullptr, "null params");
    VMS_CHECK(b->dz == nullptr || b->dz == a->dz, "dual: the gradient of z is delivered in a->dz (b->dz must be NULL or the same tensor)");
    VMS_CHECK(!b->dz_accumulate || b->dz != nullptr, "dual: b->dz_accumulate needs b->dz");
    if (vms_scan_bwd_dual_fused(a, b)) return launch_scan_bwd_pair_dual(*a, *b, static_cast<hipStream_t>(stream));
    if (int rc = vms_selective_scan_bwd(a, stream)) return rc;
    vms_scan_bwd_params b2 = *b;
    if (b2.f.z) {
        VMS_CHECK(a->dz != nullptr && a->f.z != nullptr, "dual: a->dz is required when z is given");
        b2.dz = a->dz;
        b2.dz_batch_stride = a->dz_batch_stride;
        b2.dz_d_stride = a->dz_d_stride;
        b2.dz_accumulate = 1;
    }
    return vms_selective_scan_bwd(&b2, stream);
}
