// selective_scan_fwd.hip -- forward selective scan for gfx950 (wave64).
//
// Replaces selective_scan_fwd_kernel (mamba/csrc/selective_scan/selective_scan_fwd_kernel.cuh:
// 67-303) and its launcher (:305-345).  Design (DESIGN.md "scan forward"):
//   * one WAVE owns one (batch, dim) row; a 256-thread workgroup = 4 rows of the same batch.
//     No __syncthreads anywhere: the scan is carried entirely by wave64 DPP primitives.
//   * the row is walked in chunks of 64*K elements; lane j owns the K consecutive elements
//     [j*K, (j+1)*K) of the chunk (16/32-byte vector loads per lane).
//   * per state n: serial in-register scan of the lane's K elements -> 64-lane DPP scan of
//     the lane aggregates -> seeded second pass that also contracts with C.  a = exp2(delta *
//     A * log2e) is evaluated once and kept in registers between the two passes.
//   * the running state of the 16.. dstate recurrences lives in LDS (one float per state per
//     wave, wave-private), so dstate is a runtime value (<= 256 as in the reference).
//   * checkpoints x[b,d,c,:] are written every 1024 elements (vms_hip.h).
#include <stdarg.h>
#include <stdlib.h>
#include <stdio.h>

#include "vms_common.h"

namespace vms {

constexpr int kRowsPerWG = 4;

template <typename T, int K, bool VB, bool VC, bool HZ, bool VEC>
__global__ __launch_bounds__(kRowsPerWG* kWave) void scan_fwd_kernel(const vms_scan_fwd_params p) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tiles = (p.dim + kRowsPerWG - 1) / kRowsPerWG;
    const int b = blockIdx.x / tiles;
    const int d = (blockIdx.x - b * tiles) * kRowsPerWG + wave;
    if (d >= p.dim) return;
    const int g = d / (p.dim / p.n_groups);
    const int L = p.seqlen, N = p.dstate;
    const bool rev = p.reverse != 0;
    constexpr int CS = kWave * K;  // elements per wave-chunk

    volatile lds_f32* h = (lds_f32*)smem + wave * N;  // running state per n (wave-private)
    for (int n = lane; n < N; n += kWave) h[n] = 0.f;
    const int64_t xpitch = p.x_chunk_stride ? p.x_chunk_stride : 2 * N;

    const T* u = static_cast<const T*>(p.u) + (int64_t)b * p.u_batch_stride + (int64_t)d * p.u_d_stride;
    const T* dt = static_cast<const T*>(p.delta) + (int64_t)b * p.delta_batch_stride + (int64_t)d * p.delta_d_stride;
    T* out = static_cast<T*>(p.out) + (int64_t)b * p.out_batch_stride + (int64_t)d * p.out_d_stride;
    const T* z = HZ ? static_cast<const T*>(p.z) + (int64_t)b * p.z_batch_stride + (int64_t)d * p.z_d_stride : nullptr;
    T* out_z = HZ ? static_cast<T*>(p.out_z) + (int64_t)b * p.out_z_batch_stride + (int64_t)d * p.out_z_d_stride : nullptr;
    const float* A = static_cast<const float*>(p.A) + (int64_t)d * p.A_d_stride;
    const T* Bv = VB ? static_cast<const T*>(p.B) + (int64_t)b * p.B_batch_stride + (int64_t)g * p.B_group_stride : nullptr;
    const T* Cv = VC ? static_cast<const T*>(p.C) + (int64_t)b * p.C_batch_stride + (int64_t)g * p.C_group_stride : nullptr;
    const float* Bc = !VB ? static_cast<const float*>(p.B) + (int64_t)d * p.B_d_stride : nullptr;
    const float* Cc = !VC ? static_cast<const float*>(p.C) + (int64_t)d * p.C_d_stride : nullptr;
    float* xck = static_cast<float*>(p.x) + ((int64_t)b * p.dim + d) * p.n_chunks * xpitch;
    const float Dd = p.D ? static_cast<const float*>(p.D)[d] : 0.f;
    const float bias = p.delta_bias ? static_cast<const float*>(p.delta_bias)[d] : 0.f;

    const int n_kchunks = (L + CS - 1) / CS;
    for (int c = 0; c < n_kchunks; ++c) {
        const int l0 = c * CS + lane * K;
        const int nv = L - l0;  // valid elements from l0 (may be <= 0 or >= K)
        float uv[K], dl[K], du[K], y[K];
        load_dir<T, K, VEC>(u, l0, L, rev, uv);
        load_dir<T, K, VEC>(dt, l0, L, rev, dl);
#pragma unroll
        for (int i = 0; i < K; ++i) {
            float t = dl[i] + bias;
            if (p.delta_softplus) t = softplusf_(t);
            // positions past the end must be the identity (a=1, b=0): delta = 0 does both
            dl[i] = i < nv ? t : 0.f;
            du[i] = dl[i] * uv[i];
            y[i] = Dd * uv[i];
        }
        for (int n = 0; n < N; ++n) {
            const float An = A[n * p.A_dstate_stride] * kLog2e;
            float Bn[K], Cn[K];
            if (VB) load_dir<T, K, VEC>(Bv + (int64_t)n * p.B_dstate_stride, l0, L, rev, Bn);
            if (VC) load_dir<T, K, VEC>(Cv + (int64_t)n * p.C_dstate_stride, l0, L, rev, Cn);
            const float bconst = VB ? 1.f : Bc[n * p.B_dstate_stride];
            const float cconst = VC ? 1.f : Cc[n * p.C_dstate_stride];
            float a[K], bx[K];
            float pa = 1.f, px = 0.f;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                a[i] = fast_exp2(dl[i] * An);
                bx[i] = VB ? du[i] * Bn[i] : du[i] * bconst;
                px = fmaf(a[i], px, bx[i]);
                pa *= a[i];
            }
            wave_scan_inclusive(pa, px);
            // exclusive prefix of this lane, then seed with the state carried from earlier chunks
            const float ea = dpp_mov<DPP_WAVE_SHR1, 0xf>(1.f, pa);
            const float ex = dpp_mov<DPP_WAVE_SHR1, 0xf>(0.f, px);
            const float hin = h[n];
            float xs = fmaf(ea, hin, ex);
            const float hout = fmaf(readlane_f(pa, 63), hin, readlane_f(px, 63));
            if (lane == 0) h[n] = hout;
            if (p.x_has_sub == 1 && (lane & 7) == 7) {
                // 128-element sub-checkpoints for the backward kernel: the state after this lane's
                // last element, kept by every 8th lane (8 lanes x 16 elements = 128)
                static_assert(CS == 1024 && K == 16, "sub-checkpoint indexing assumes 64 x 16 wave chunks");
                xck[(int64_t)(c >> 1) * xpitch + 2 * N + ((c & 1) * 8 + (lane >> 3)) * N + n] = fmaf(pa, hin, px);
            }
            float x_mid = 0.f;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                xs = fmaf(a[i], xs, bx[i]);
                y[i] = fmaf(VC ? Cn[i] : cconst, xs, y[i]);
                if (i == K / 2 - 1) x_mid = xs;
            }
            // 8-element checkpoints (vms_hip.h, x_has_sub == 3): the state after the lane's 8th and 16th element
            if (p.x_has_sub == 3 && nv > 0) {
                float* const xq = xck + (int64_t)(c >> 1) * xpitch + 2 * N + (((n >> 2) * 256 + (c & 1) * 128 + 2 * lane) * 4 + (n & 3));
                xq[0] = x_mid;
                xq[4] = xs;
            }
        }
        store_dir<T, K, VEC>(out, l0, L, rev, y);
        if (HZ) {
            float zv[K];
            load_dir<T, K, VEC>(z, l0, L, rev, zv);
#pragma unroll
            for (int i = 0; i < K; ++i) y[i] *= zv[i] * sigmoidf_(zv[i]);
            if (p.out_z_accumulate) {  // out_z += (vms_hip.h)
                float old[K];
                load_dir<T, K, VEC>(out_z, l0, L, rev, old);
#pragma unroll
                for (int i = 0; i < K; ++i) y[i] += old[i];
            }
            store_dir<T, K, VEC>(out_z, l0, L, rev, y);
        }
        // checkpoints every 1024 elements (and at the end of the sequence)
        const int pos = (c + 1) * CS;
        const bool last = c == n_kchunks - 1;
        if ((pos & 1023) == 0 || last) {
            const int blk = last ? (L - 1) / 2048 : (pos - 1) / 2048;
            const int r = (last ? L : pos) - blk * 2048;  // elements of block blk seen so far
            float* xb = xck + (int64_t)blk * xpitch;
            const bool w_even = r <= 1024, w_odd = r == 2048 || last;
            for (int n = lane; n < N; n += kWave) {
                const float s = h[n];
                if (w_even) xb[2 * n] = s;
                if (w_odd) xb[2 * n + 1] = s;
            }
        }
    }
}

template <typename T, int K, bool VB, bool VC, bool HZ>
static int launch_fwd(const vms_scan_fwd_params& p, bool vec, hipStream_t stream) {
    const int tiles = (p.dim + kRowsPerWG - 1) / kRowsPerWG;
    dim3 grid(p.batch * tiles), block(kRowsPerWG * kWave);
    const size_t smem = sizeof(float) * kRowsPerWG * p.dstate;
    if (vec)
        hipLaunchKernelGGL((scan_fwd_kernel<T, K, VB, VC, HZ, true>), grid, block, smem, stream, p);
    else
        hipLaunchKernelGGL((scan_fwd_kernel<T, K, VB, VC, HZ, false>), grid, block, smem, stream, p);
    VMS_LAUNCH_CHECK();
    return VMS_OK;
}

template <typename T, int K>
static int dispatch_fwd(const vms_scan_fwd_params& p, bool vec, hipStream_t s) {
    const bool vb = p.is_variable_B, vc = p.is_variable_C, hz = p.z != nullptr;
#define VMS_CASE(B_, C_, Z_) \
    if (vb == B_ && vc == C_ && hz == Z_) return launch_fwd<T, K, B_, C_, Z_>(p, vec, s);
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

int validate_scan_common(const vms_scan_fwd_params& p) {
    VMS_CHECK(p.dtype == VMS_F32 || p.dtype == VMS_F16 || p.dtype == VMS_BF16, "input dtype must be fp32/fp16/bf16");
    VMS_CHECK(p.batch > 0 && p.dim > 0 && p.seqlen > 0 && p.dstate > 0, "empty problem");
    VMS_CHECK(p.dstate <= 256, "selective_scan only supports state dimension <= 256");
    VMS_CHECK(p.n_groups >= 1 && p.dim % p.n_groups == 0, "dim must be divisible by n_groups");
    VMS_CHECK(p.n_chunks == (p.seqlen + 2047) / 2048, "n_chunks must be ceil(seqlen / 2048)");
    VMS_CHECK(p.u && p.delta && p.A && p.B && p.C, "u, delta, A, B, C are required");
    VMS_CHECK(p.x_has_sub >= 0 && p.x_has_sub <= 3, "x_has_sub out of range");
    if (p.is_complex) return VMS_OK;   // the complex layouts: validate_complex, selective_scan_complex.hip
    VMS_CHECK(p.x_has_sub != 1 || p.x_chunk_stride >= 18 * (int64_t)p.dstate, "x_has_sub == 1 needs an x pitch >= 18 * dstate");
    VMS_CHECK(p.x_has_sub != 3 || (p.x_chunk_stride >= 258 * (int64_t)p.dstate && p.x_chunk_stride % 2 == 0 &&
                                   (reinterpret_cast<uintptr_t>(p.x) & 7) == 0),
              "x_has_sub == 3 needs an 8-byte aligned x with an even pitch >= 258 * dstate");
    return VMS_OK;
}

bool scan_fwd_pair_eligible(const vms_scan_fwd_params& p, bool vec);
bool scan_bwd_pair_lane_ckpt_ok(const vms_scan_fwd_params& p);   // selective_scan_bwd_pair.hip
int scan_fwd_pair_segments(const vms_scan_fwd_params& p);
int64_t scan_fwd_pair_ws_bytes(const vms_scan_fwd_params& p);
int launch_scan_fwd_pair(const vms_scan_fwd_params& p, hipStream_t stream);

int launch_scan_fwd_complex(const vms_scan_fwd_params& p, bool vec, hipStream_t stream);   // selective_scan_complex.hip
bool scan_short_eligible(const vms_scan_fwd_params& p);                                      // selective_scan_short.hip
bool scan_short_takes(const vms_scan_fwd_params& p);
bool scan_short_takes_shape(const vms_scan_fwd_params& p);
int64_t scan_short_x_pitch(const vms_scan_fwd_params& p);
int launch_scan_fwd_short(const vms_scan_fwd_params& p, hipStream_t stream);

bool scan_fwd_vec_ok(const vms_scan_fwd_params& p) {
    const int es = p.dtype == VMS_F32 ? 4 : 2;
    bool ok = aligned16(p.u) && aligned16(p.delta) && aligned16(p.out) && mult16(p.u_batch_stride, es) &&
              mult16(p.u_d_stride, es) && mult16(p.delta_batch_stride, es) && mult16(p.delta_d_stride, es) &&
              mult16(p.out_batch_stride, es) && mult16(p.out_d_stride, es);
    if (p.z)
        ok = ok && aligned16(p.z) && aligned16(p.out_z) && mult16(p.z_batch_stride, es) &&
             mult16(p.z_d_stride, es) && mult16(p.out_z_batch_stride, es) && mult16(p.out_z_d_stride, es);
    if (p.is_variable_B)
        ok = ok && aligned16(p.B) && mult16(p.B_batch_stride, es) && mult16(p.B_group_stride, es) &&
             mult16(p.B_dstate_stride, es);
    if (p.is_variable_C)
        ok = ok && aligned16(p.C) && mult16(p.C_batch_stride, es) && mult16(p.C_group_stride, es) &&
             mult16(p.C_dstate_stride, es);
    return ok;
}

}  // namespace vms

using namespace vms;

extern "C" int64_t vms_scan_fwd_workspace_bytes(const vms_scan_fwd_params* p);
// reverse_from (ABI v5): the two sub-batches as two problems.  Every kernel generation serves it this way; the results are
// by construction those of two calls.  The workspace is split in the same proportion vms_scan_fwd_workspace_bytes reports.
static inline const void* off_c(const void* ptr, int64_t elems, int es) { return ptr ? static_cast<const char*>(ptr) + elems * es : nullptr; }
static inline void* off_m(void* ptr, int64_t elems, int es) { return ptr ? static_cast<char*>(ptr) + elems * es : nullptr; }
static inline int64_t round256(int64_t n) { return (n + 255) & ~(int64_t)255; }

void vms::scan_fwd_sub_batches(const vms_scan_fwd_params& p, vms_scan_fwd_params& lo, vms_scan_fwd_params& hi) {
    const int rf = p.reverse_from, es = p.dtype == VMS_F32 ? 4 : 2;
    lo = p; lo.batch = rf; lo.reverse_from = 0; lo.reverse = 0;
    hi = p; hi.batch = p.batch - rf; hi.reverse_from = 0; hi.reverse = 1;
    hi.u = off_c(p.u, rf * p.u_batch_stride, es);
    hi.delta = off_c(p.delta, rf * p.delta_batch_stride, es);
    hi.z = off_c(p.z, rf * p.z_batch_stride, es);
    hi.out = off_m(p.out, rf * p.out_batch_stride, es);
    hi.out_z = off_m(p.out_z, rf * p.out_z_batch_stride, es);
    if (p.is_variable_B) hi.B = off_c(p.B, rf * p.B_batch_stride, es);
    if (p.is_variable_C) hi.C = off_c(p.C, rf * p.C_batch_stride, es);
    const int64_t xpitch = p.x_chunk_stride ? p.x_chunk_stride : 2 * p.dstate;
    hi.x = off_m(p.x, (int64_t)rf * p.dim * p.n_chunks * xpitch, 4);
    lo.workspace = hi.workspace = nullptr;
    lo.workspace_bytes = hi.workspace_bytes = 0;
}

static int scan_fwd_mixed(const vms_scan_fwd_params& p, void* stream) {
    vms_scan_fwd_params lo, hi;
    scan_fwd_sub_batches(p, lo, hi);
    const int64_t wl = round256(vms_scan_fwd_workspace_bytes(&lo)), wh = vms_scan_fwd_workspace_bytes(&hi);
    if (p.workspace != nullptr && p.workspace_bytes >= wl + wh) {
        if (wl > 0) { lo.workspace = p.workspace; lo.workspace_bytes = wl; }
        if (wh > 0) { hi.workspace = static_cast<char*>(p.workspace) + wl; hi.workspace_bytes = wh; }
    }
    if (int rc = vms_selective_scan_fwd(&lo, stream)) return rc;
    return vms_selective_scan_fwd(&hi, stream);
}

extern "C" int vms_selective_scan_fwd(const vms_scan_fwd_params* pp, void* stream) {
    VMS_CHECK(pp != nullptr, "null params");
    const vms_scan_fwd_params& p = *pp;
    if (int rc = validate_scan_common(p)) return rc;
    if (p.reverse_from != 0) {
        VMS_CHECK(p.reverse_from > 0 && p.reverse_from <= p.batch && p.reverse == 0, "reverse_from must be in (0, batch] with reverse == 0");
        VMS_CHECK(p.x_has_sub != 2, "reverse_from is not available with the rows layout");
        if (p.reverse_from < p.batch) {
            // one launch when the paired LDS kernel takes the problem as it is; otherwise the two sub-batches as two problems
            // (the complex kernels take the direction per batch entry as it is)
            const bool native = p.is_complex || (scan_impl_level(p) >= VMS_IMPL_PAIR && scan_fwd_pair_eligible(p, scan_fwd_vec_ok(p)) &&
                                                 scan_fwd_pair_native_mixed(p) && !scan_short_takes(p));
            if (!native) return scan_fwd_mixed(p, stream);
        }
    }
    VMS_CHECK(p.out && p.x, "out and x must be provided by the caller");
    VMS_CHECK((p.z == nullptr) == (p.out_z == nullptr), "out_z must be given iff z is given");
    VMS_CHECK(!p.out_z_accumulate || p.z != nullptr, "out_z_accumulate needs z / out_z");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool vec = scan_fwd_vec_ok(p);
    if (p.is_complex) return launch_scan_fwd_complex(p, vec, s);
    VMS_CHECK(scan_impl_valid(p.impl) && p.segments >= 0, "impl / segments out of range");
    const int level = scan_impl_level(p);
    VMS_CHECK(p.x_has_sub != 2, "x_has_sub == 2 (the row-major layout of rounds 1-3) is no longer built");
    // short rows, many of them (TimeMamba's scans along time: seqlen 4 ... 16, batch x 196 rows per channel): a lane per row
    if (level >= VMS_IMPL_PAIR && p.x_has_sub == 0 && scan_short_eligible(p)) return launch_scan_fwd_short(p, s);
    if (level >= VMS_IMPL_PAIR && scan_fwd_pair_eligible(p, vec)) return launch_scan_fwd_pair(p, s);
    set_last_kernel("scan_fwd_generic");
    switch (p.dtype) {
        case VMS_F32: return dispatch_fwd<float, 16>(p, vec, s);
        case VMS_F16: return dispatch_fwd<f16_t, 16>(p, vec, s);
        default: return dispatch_fwd<bf16_t, 16>(p, vec, s);
    }
}

extern "C" int64_t vms_scan_fwd_workspace_bytes(const vms_scan_fwd_params* p) {
    if (p == nullptr || p->is_complex) return 0;
    if (p->reverse_from > 0 && p->reverse_from < p->batch) {
        if (scan_impl_level(*p) >= VMS_IMPL_PAIR && scan_fwd_pair_eligible(*p, true) && scan_fwd_pair_native_mixed(*p) && !scan_short_takes(*p)) return 0;
        vms_scan_fwd_params lo, hi;
        scan_fwd_sub_batches(*p, lo, hi);
        const int64_t wl = vms_scan_fwd_workspace_bytes(&lo), wh = vms_scan_fwd_workspace_bytes(&hi);
        return wl + wh > 0 ? round256(wl) + wh : 0;
    }
    const int level = scan_impl_level(*p);
    // the paired kernel's (P, q) state carries when it wants to split long rows into ranges
    if (level >= VMS_IMPL_PAIR && scan_fwd_pair_eligible(*p, true) && scan_fwd_pair_segments(*p) > 1)
        return scan_fwd_pair_ws_bytes(*p);
    return 0;
}
extern "C" int64_t vms_scan_x_pitch(const vms_scan_fwd_params* pp, int32_t mode) {
    if (pp == nullptr || pp->dstate <= 0) return 0;
    const vms_scan_fwd_params& p = *pp;
    if (p.is_complex) return (mode == 0 ? 2 : 6) * (int64_t)p.dstate;   // complex elements (vms_hip.h is_complex)
    if (mode == 0) return 2 * (int64_t)p.dstate;
    // the lane-per-row kernels of short sequences keep no checkpoints: the reference's x and nothing behind it (the 8-element
    // layout costs 16.5 KB per row whatever its length: 20 GB per scan at (1568, 8, 768))
    // (rows of 17 .. 64 elements run as segments of 16 chained through x: 2 N + (segments - 1) N, selective_scan_short.hip)
    if (vms::scan_short_takes_shape(p)) return vms::scan_short_x_pitch(p);
    if ((mode == 3 || mode == -1) && vms::scan_bwd_pair_lane_ckpt_ok(p)) return 258 * (int64_t)p.dstate;
    return 18 * (int64_t)p.dstate;
}

extern "C" int64_t vms_scan_x_elems(const vms_scan_fwd_params* p) {
    if (p == nullptr) return 0;
    const int64_t ref = (int64_t)p->batch * p->dim * p->n_chunks * 2 * p->dstate;
    return ref;
}
