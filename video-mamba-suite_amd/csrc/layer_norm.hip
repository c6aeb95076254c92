// layer_norm.hip -- fused residual-add + LayerNorm / RMSNorm forward and backward for gfx950 (wave64).
//
// Replaces the Triton kernels of mamba/mamba_ssm/ops/triton/layernorm.py (_layer_norm_fwd_1pass_kernel :51-120,
// _layer_norm_bwd_kernel :176-288) behind layer_norm_fn / rms_norm_fn / RMSNorm -- the op that sits in front
// of every Mamba mixer in ViViM / LSTR / UniVTG.  Semantics = layer_norm_ref / rms_norm_ref
// with upcast (:19-48): fp32 statistics on s = x + residual, the sum optionally written out (prenorm /
// residual_in_fp32), y = (s - mean) * rstd * w + b.
//
// Pure HBM streaming with two row reductions.  One WAVE per row, rows walked persistently (grid = a few
// waves per SIMD), a lane owns the 16-byte pieces lane, lane + 64, ... of the row and keeps them in
// registers between the reductions (K pieces per lane, K in {1, 2, 4, 8}: rows up to 4096 16-bit or 2048
// fp32 elements); reductions are DPP wave sums (no LDS, no barriers).  Rows that do not fit or are not
// 16-byte friendly take a three-pass element-wise kernel.  Backward: dw / db are accumulated per wave in
// registers over the rows it walks and written as one partial row per wave; the caller sums the partials
// (the reference does the same with one partial per SM, layernorm.py:316-375).
// Weight and bias are fp32 (the Python layer widens them: `cols` elements).
#include "vms_common.h"

namespace vms {

constexpr int kNormWaves = 4;  // waves (rows in flight) per workgroup

template <typename T, int E>
__device__ __forceinline__ void ld_chunk(const T* p, float (&o)[E]) {
    constexpr int EPV = 16 / sizeof(T);
#pragma unroll
    for (int v = 0; v < E / EPV; ++v) {
        const vec_t<T, EPV> t = reinterpret_cast<const vec_t<T, EPV>*>(p)[v];
#pragma unroll
        for (int e = 0; e < EPV; ++e) o[v * EPV + e] = static_cast<float>(t[e]);
    }
}
template <typename T, int E>
__device__ __forceinline__ void st_chunk(T* p, const float (&o)[E]) {
    constexpr int EPV = 16 / sizeof(T);
#pragma unroll
    for (int v = 0; v < E / EPV; ++v) {
        vec_t<T, EPV> t;
#pragma unroll
        for (int e = 0; e < EPV; ++e) t[e] = static_cast<T>(o[v * EPV + e]);
        reinterpret_cast<vec_t<T, EPV>*>(p)[v] = t;
    }
}

// ---- forward, register resident ------------------------------------------------------------------------
template <typename TX, typename TS, bool RMS, int K>
__global__ __launch_bounds__(kNormWaves* kWave) void norm_fwd_vec_kernel(const vms_norm_params p) {
    constexpr int E = 16 / sizeof(TX);  // elements per piece
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int N = p.cols;
    const float inv_n = 1.f / N;
    const float* w = static_cast<const float*>(p.weight);
    const float* bs = static_cast<const float*>(p.bias);
    for (int64_t row = (int64_t)blockIdx.x * kNormWaves + wave; row < p.rows; row += (int64_t)gridDim.x * kNormWaves) {
        const TX* x = static_cast<const TX*>(p.x) + row * p.x_row_stride;
        const TS* res = p.residual ? static_cast<const TS*>(p.residual) + row * p.residual_row_stride : nullptr;
        TS* ro = p.residual_out ? static_cast<TS*>(p.residual_out) + row * p.residual_out_row_stride : nullptr;
        TX* y = static_cast<TX*>(p.y) + row * p.y_row_stride;
        float s[K][E];
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int c0 = (lane + 64 * k) * E;
            if (c0 < N) {
                ld_chunk<TX, E>(x + c0, s[k]);
                if (res) {
                    float r[E];
                    ld_chunk<TS, E>(res + c0, r);
#pragma unroll
                    for (int e = 0; e < E; ++e) s[k][e] += r[e];
                }
                if (ro) st_chunk<TS, E>(ro + c0, s[k]);
#pragma unroll
                for (int e = 0; e < E; ++e) sum += s[k][e];
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e) s[k][e] = 0.f;
            }
        }
        const float mean = RMS ? 0.f : wave_sum(sum) * inv_n;
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int c0 = (lane + 64 * k) * E;
            if (c0 < N) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const float d = s[k][e] - mean;
                    sq = fmaf(d, d, sq);
                }
            }
        }
        const float rstd = 1.f / sqrtf(wave_sum(sq) * inv_n + p.eps);
        if (lane == 0) {
            if (!RMS && p.mean) p.mean[row] = mean;
            p.rstd[row] = rstd;
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int c0 = (lane + 64 * k) * E;
            if (c0 < N) {
                float wv[E], o[E];
                ld_chunk<float, E>(w + c0, wv);
#pragma unroll
                for (int e = 0; e < E; ++e) o[e] = (s[k][e] - mean) * rstd * wv[e];
                if (bs) {
                    float bv[E];
                    ld_chunk<float, E>(bs + c0, bv);
#pragma unroll
                    for (int e = 0; e < E; ++e) o[e] += bv[e];
                }
                st_chunk<TX, E>(y + c0, o);
            }
        }
    }
}

// ---- forward, any row length / alignment: three passes over the row ---------------------------------------
template <typename TX, typename TS, bool RMS>
__global__ __launch_bounds__(kNormWaves* kWave) void norm_fwd_gen_kernel(const vms_norm_params p) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int N = p.cols;
    const float inv_n = 1.f / N;
    const float* w = static_cast<const float*>(p.weight);
    const float* bs = static_cast<const float*>(p.bias);
    for (int64_t row = (int64_t)blockIdx.x * kNormWaves + wave; row < p.rows; row += (int64_t)gridDim.x * kNormWaves) {
        const TX* x = static_cast<const TX*>(p.x) + row * p.x_row_stride;
        const TS* res = p.residual ? static_cast<const TS*>(p.residual) + row * p.residual_row_stride : nullptr;
        TS* ro = p.residual_out ? static_cast<TS*>(p.residual_out) + row * p.residual_out_row_stride : nullptr;
        TX* y = static_cast<TX*>(p.y) + row * p.y_row_stride;
        auto at = [&](int c) { return static_cast<float>(x[c]) + (res ? static_cast<float>(res[c]) : 0.f); };
        float sum = 0.f;
        for (int c = lane; c < N; c += 64) sum += at(c);
        const float mean = RMS ? 0.f : wave_sum(sum) * inv_n;
        float sq = 0.f;
        for (int c = lane; c < N; c += 64) {
            const float d = at(c) - mean;
            sq = fmaf(d, d, sq);
        }
        const float rstd = 1.f / sqrtf(wave_sum(sq) * inv_n + p.eps);
        if (lane == 0) {
            if (!RMS && p.mean) p.mean[row] = mean;
            p.rstd[row] = rstd;
        }
        for (int c = lane; c < N; c += 64) {
            const float s = at(c);
            if (ro) ro[c] = static_cast<TS>(s);
            y[c] = static_cast<TX>((s - mean) * rstd * w[c] + (bs ? bs[c] : 0.f));
        }
    }
}

// ---- backward, register resident ----------------------------------------------------------------------------
// s = saved pre-norm sum (dtype TS), dy (TX), optional dres_out (TS) -> dx (TX), optional dres_in (TS),
// dw_partial / db_partial [gridDim.x][cols] fp32 (one row per workgroup, always fully written)
template <typename TX, typename TS, bool RMS, int K>
__global__ __launch_bounds__(kNormWaves* kWave) void norm_bwd_vec_kernel(const vms_norm_bwd_params q) {
    const vms_norm_params& p = q.f;
    constexpr int E = 16 / sizeof(TX);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int N = p.cols;
    const float inv_n = 1.f / N;
    const float* w = static_cast<const float*>(p.weight);
    float wv[K][E], dwa[K][E], dba[K][E];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int c0 = (lane + 64 * k) * E;
#pragma unroll
        for (int e = 0; e < E; ++e) { wv[k][e] = 0.f; dwa[k][e] = 0.f; dba[k][e] = 0.f; }
        if (c0 < N) ld_chunk<float, E>(w + c0, wv[k]);
    }
    for (int64_t row = (int64_t)blockIdx.x * kNormWaves + wave; row < p.rows; row += (int64_t)gridDim.x * kNormWaves) {
        const TS* s = static_cast<const TS*>(q.s) + row * q.s_row_stride;
        const TX* dy = static_cast<const TX*>(q.dy) + row * q.dy_row_stride;
        const TS* dro = q.dres_out ? static_cast<const TS*>(q.dres_out) + row * q.dres_out_row_stride : nullptr;
        TX* dx = static_cast<TX*>(q.dx) + row * q.dx_row_stride;
        TS* dri = q.dres_in ? static_cast<TS*>(q.dres_in) + row * q.dres_in_row_stride : nullptr;
        const float mean = (RMS || !p.mean) ? 0.f : p.mean[row];
        const float rstd = p.rstd[row];
        float xh[K][E], wdy[K][E];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int c0 = (lane + 64 * k) * E;
            if (c0 < N) {
                float sv[E], g[E];
                ld_chunk<TS, E>(s + c0, sv);
                ld_chunk<TX, E>(dy + c0, g);
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    xh[k][e] = (sv[e] - mean) * rstd;
                    wdy[k][e] = wv[k][e] * g[e];
                    c1 = fmaf(xh[k][e], wdy[k][e], c1);
                    c2 += wdy[k][e];
                    dwa[k][e] = fmaf(g[e], xh[k][e], dwa[k][e]);
                    dba[k][e] += g[e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e) { xh[k][e] = 0.f; wdy[k][e] = 0.f; }
            }
        }
        c1 = wave_sum(c1) * inv_n;
        c2 = RMS ? 0.f : wave_sum(c2) * inv_n;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int c0 = (lane + 64 * k) * E;
            if (c0 < N) {
                float o[E];
#pragma unroll
                for (int e = 0; e < E; ++e) o[e] = (wdy[k][e] - (xh[k][e] * c1 + c2)) * rstd;
                if (dro) {
                    float r[E];
                    ld_chunk<TS, E>(dro + c0, r);
#pragma unroll
                    for (int e = 0; e < E; ++e) o[e] += r[e];
                }
                if (dri) st_chunk<TS, E>(dri + c0, o);
                st_chunk<TX, E>(dx + c0, o);
            }
        }
    }
    // the workgroup's four per-wave sums become ONE partial row (summed through LDS: a quarter of the partial bytes to write here
    // and to sum afterwards -- (8, 3136, 768): 8,192 rows x 768 floats = 25 MB per array, more than a third of the kernel's traffic)
    __shared__ float red[(kNormWaves - 1) * K * E * 64];
    const int64_t prow = blockIdx.x;
    auto emit = [&](float (&acc)[K][E], float* dst) __attribute__((always_inline)) {
        if (wave > 0) {
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int e = 0; e < E; ++e) red[((wave - 1) * K * E + k * E + e) * 64 + lane] = acc[k][e];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int c0 = (lane + 64 * k) * E;
                float o[E];
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    o[e] = acc[k][e];
#pragma unroll
                    for (int w2 = 0; w2 < kNormWaves - 1; ++w2) o[e] += red[(w2 * K * E + k * E + e) * 64 + lane];
                }
                if (c0 < N) st_chunk<float, E>(dst + prow * N + c0, o);
            }
        }
    };
    emit(dwa, q.dw_partial);
    if (q.db_partial) {      // (uniform)
        __syncthreads();     // the first pass's reads of `red` are done
        emit(dba, q.db_partial);
    }
}

template <typename TX, typename TS, bool RMS>
__global__ __launch_bounds__(kWave) void norm_bwd_gen_kernel(const vms_norm_bwd_params q) {
    const vms_norm_params& p = q.f;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int N = p.cols;
    const float inv_n = 1.f / N;
    const float* w = static_cast<const float*>(p.weight);
    const int64_t prow = blockIdx.x;     // launched with ONE wave per workgroup: a partial row per workgroup, as the vector kernel leaves
    (void)wave;
    for (int c = lane; c < N; c += 64) {
        q.dw_partial[prow * N + c] = 0.f;
        if (q.db_partial) q.db_partial[prow * N + c] = 0.f;
    }
    for (int64_t row = prow; row < p.rows; row += (int64_t)gridDim.x) {
        const TS* s = static_cast<const TS*>(q.s) + row * q.s_row_stride;
        const TX* dy = static_cast<const TX*>(q.dy) + row * q.dy_row_stride;
        const TS* dro = q.dres_out ? static_cast<const TS*>(q.dres_out) + row * q.dres_out_row_stride : nullptr;
        TX* dx = static_cast<TX*>(q.dx) + row * q.dx_row_stride;
        TS* dri = q.dres_in ? static_cast<TS*>(q.dres_in) + row * q.dres_in_row_stride : nullptr;
        const float mean = (RMS || !p.mean) ? 0.f : p.mean[row];
        const float rstd = p.rstd[row];
        float c1 = 0.f, c2 = 0.f;
        for (int c = lane; c < N; c += 64) {
            const float xh = (static_cast<float>(s[c]) - mean) * rstd, g = static_cast<float>(dy[c]);
            const float wdy = w[c] * g;
            c1 = fmaf(xh, wdy, c1);
            c2 += wdy;
            q.dw_partial[prow * N + c] += g * xh;  // column c belongs to this lane only
            if (q.db_partial) q.db_partial[prow * N + c] += g;
        }
        c1 = wave_sum(c1) * inv_n;
        c2 = RMS ? 0.f : wave_sum(c2) * inv_n;
        for (int c = lane; c < N; c += 64) {
            const float xh = (static_cast<float>(s[c]) - mean) * rstd, wdy = w[c] * static_cast<float>(dy[c]);
            float o = (wdy - (xh * c1 + c2)) * rstd;
            if (dro) o += static_cast<float>(dro[c]);
            if (dri) dri[c] = static_cast<TS>(o);
            dx[c] = static_cast<TX>(o);
        }
    }
}

// ---- host side ----------------------------------------------------------------------------------------------
static int norm_grid(int64_t rows) {
    const int64_t want = (rows + kNormWaves - 1) / kNormWaves;
    return (int)(want < 2048 ? want : 2048);  // 8192 waves: 8 per SIMD, rows walked persistently
}

static int validate_norm(const vms_norm_params& p) {
    VMS_CHECK(p.x_dtype == VMS_F32 || p.x_dtype == VMS_F16 || p.x_dtype == VMS_BF16, "x dtype must be fp32/fp16/bf16");
    VMS_CHECK(p.res_dtype == VMS_F32 || p.res_dtype == VMS_F16 || p.res_dtype == VMS_BF16, "residual dtype must be fp32/fp16/bf16");
    VMS_CHECK(p.rows > 0 && p.cols > 0, "empty problem");
    VMS_CHECK(p.weight != nullptr && p.rstd != nullptr, "weight and rstd are required");
    return VMS_OK;
}

static bool norm_vec_ok(const vms_norm_params& p, int esz, int ssz, const void* const* ptrs, const int64_t* strides,
                        const int* sizes, int n) {
    const int E = 16 / esz;
    if (p.cols % E != 0 || p.cols > 64 * E * 8) return false;
    for (int i = 0; i < n; ++i)
        if (ptrs[i] && (!aligned16(ptrs[i]) || ((strides[i] * sizes[i]) & 15))) return false;
    (void)ssz;
    return aligned16(p.weight) && (!p.bias || aligned16(p.bias));
}

template <typename TX, typename TS, bool RMS>
static int norm_fwd_launch(const vms_norm_params& p, bool vec, hipStream_t s) {
    constexpr int E = 16 / sizeof(TX);
    dim3 grid(norm_grid(p.rows)), block(kNormWaves * kWave);
    const int pieces = (p.cols / E + 63) / 64;
    if (!vec) hipLaunchKernelGGL((norm_fwd_gen_kernel<TX, TS, RMS>), grid, block, 0, s, p);
    else if (pieces <= 1) hipLaunchKernelGGL((norm_fwd_vec_kernel<TX, TS, RMS, 1>), grid, block, 0, s, p);
    else if (pieces <= 2) hipLaunchKernelGGL((norm_fwd_vec_kernel<TX, TS, RMS, 2>), grid, block, 0, s, p);
    else if (pieces <= 4) hipLaunchKernelGGL((norm_fwd_vec_kernel<TX, TS, RMS, 4>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((norm_fwd_vec_kernel<TX, TS, RMS, 8>), grid, block, 0, s, p);
    VMS_LAUNCH_CHECK();
    return VMS_OK;
}
template <typename TX, typename TS, bool RMS>
static int norm_bwd_launch(const vms_norm_bwd_params& q, bool vec, hipStream_t s) {
    constexpr int E = 16 / sizeof(TX);
    const vms_norm_params& p = q.f;
    dim3 grid(q.n_partials), block(kNormWaves * kWave);
    const int pieces = (p.cols / E + 63) / 64;
    if (!vec || pieces > 4) hipLaunchKernelGGL((norm_bwd_gen_kernel<TX, TS, RMS>), grid, dim3(kWave), 0, s, q);
    else if (pieces <= 1) hipLaunchKernelGGL((norm_bwd_vec_kernel<TX, TS, RMS, 1>), grid, block, 0, s, q);
    else if (pieces <= 2) hipLaunchKernelGGL((norm_bwd_vec_kernel<TX, TS, RMS, 2>), grid, block, 0, s, q);
    else hipLaunchKernelGGL((norm_bwd_vec_kernel<TX, TS, RMS, 4>), grid, block, 0, s, q);
    VMS_LAUNCH_CHECK();
    return VMS_OK;
}

// The residual stream (residual in / residual_out, and in backward s / dres_out / dres_in) may have any of the three dtypes
// whatever x's is, as in the reference's Triton kernels (ops/triton/layernorm.py:122-173).  The pairs the suite uses -- the
// stream in x's dtype, or in fp32 (residual_in_fp32) -- run on the register-resident vector kernels; the remaining pairs
// (fp32 x with a 16-bit stream, fp16 x with a bf16 stream and the reverse) on the element-wise kernels.
#define VMS_NORM_TS(FN, TX_, vec, s)                                                                          \
    switch (ts_) {                                                                                            \
        case VMS_F32: return rms_ ? FN<TX_, float, true>(q_, vec && fast_, s) : FN<TX_, float, false>(q_, vec && fast_, s);   \
        case VMS_F16: return rms_ ? FN<TX_, f16_t, true>(q_, vec && fast_, s) : FN<TX_, f16_t, false>(q_, vec && fast_, s);   \
        default: return rms_ ? FN<TX_, bf16_t, true>(q_, vec && fast_, s) : FN<TX_, bf16_t, false>(q_, vec && fast_, s);      \
    }
#define VMS_NORM_DISPATCH(FN, ARG, vec, s)                                                                    \
    do {                                                                                                      \
        const int ts_ = (ARG).res_dtype;                                                                       \
        const bool fast_ = ts_ == (ARG).x_dtype || ts_ == VMS_F32;   /* pairs the vector kernels are written for */ \
        const bool rms_ = (ARG).is_rms != 0;                                                                   \
        switch ((ARG).x_dtype) {                                                                               \
            case VMS_F32: VMS_NORM_TS(FN, float, vec, s)                                                       \
            case VMS_F16: VMS_NORM_TS(FN, f16_t, vec, s)                                                       \
            default: VMS_NORM_TS(FN, bf16_t, vec, s)                                                           \
        }                                                                                                      \
    } while (0)

}  // namespace vms

using namespace vms;

extern "C" int vms_layer_norm_fwd(const vms_norm_params* pp, void* stream) {
    VMS_CHECK(pp != nullptr, "null params");
    const vms_norm_params& q_ = *pp;
    if (int rc = validate_norm(q_)) return rc;
    VMS_CHECK(q_.x && q_.y, "x and y are required");
    const int esz = q_.x_dtype == VMS_F32 ? 4 : 2, ssz = q_.res_dtype == VMS_F32 ? 4 : 2;
    const void* ptrs[4] = {q_.x, q_.y, q_.residual, q_.residual_out};
    const int64_t strides[4] = {q_.x_row_stride, q_.y_row_stride, q_.residual_row_stride, q_.residual_out_row_stride};
    const int sizes[4] = {esz, esz, ssz, ssz};
    const bool vec = norm_vec_ok(q_, esz, ssz, ptrs, strides, sizes, 4);
    hipStream_t s = static_cast<hipStream_t>(stream);
    VMS_NORM_DISPATCH(norm_fwd_launch, q_, vec, s);
}

extern "C" int vms_layer_norm_bwd_partials(const vms_norm_params* pp) {
    if (pp == nullptr || pp->rows <= 0) return 0;
    return norm_grid(pp->rows);     // one partial row per workgroup
}

extern "C" int vms_layer_norm_bwd(const vms_norm_bwd_params* qq, void* stream) {
    VMS_CHECK(qq != nullptr, "null params");
    const vms_norm_bwd_params& q_ = *qq;
    if (int rc = validate_norm(q_.f)) return rc;
    VMS_CHECK(q_.s && q_.dy && q_.dx && q_.dw_partial, "s, dy, dx and dw_partial are required");
    VMS_CHECK(q_.n_partials == vms_layer_norm_bwd_partials(&q_.f), "n_partials must be vms_layer_norm_bwd_partials()");
    const int esz = q_.f.x_dtype == VMS_F32 ? 4 : 2, ssz = q_.f.res_dtype == VMS_F32 ? 4 : 2;
    const void* ptrs[5] = {q_.s, q_.dy, q_.dx, q_.dres_out, q_.dres_in};
    const int64_t strides[5] = {q_.s_row_stride, q_.dy_row_stride, q_.dx_row_stride, q_.dres_out_row_stride, q_.dres_in_row_stride};
    const int sizes[5] = {ssz, esz, esz, ssz, ssz};
    const bool vec = norm_vec_ok(q_.f, esz, ssz, ptrs, strides, sizes, 5) && aligned16(q_.dw_partial) &&
                     (!q_.db_partial || aligned16(q_.db_partial)) && (q_.f.cols % 4 == 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    VMS_NORM_DISPATCH(norm_bwd_launch, q_.f, vec, s);
}

// ---- sum of the backward's partial rows ---------------------------------------------------------------------
// dw[c] = sum_p dw_partial[p][c] (and db): a workgroup owns 16 columns; thread = (row lane tid / 4, 4 columns tid % 4), 64 row
// lanes walk the partial rows 64 apart with 16-byte loads (all of a thread's loads independent), their sums meet in LDS.
// (torch's sum over the same (2048, 768) array: 14 us + a 5 us fill of its multi-block semaphores; this: one 5 us launch for both)
namespace vms {
template <typename TO>
__global__ __launch_bounds__(256) void norm_bwd_finish_kernel(const float* __restrict__ dw_p, const float* __restrict__ db_p, const int n_part,
                                                              const int cols, TO* __restrict__ dw, TO* __restrict__ db) {
    __shared__ float red[64][17];
    const float* const src = blockIdx.y ? db_p : dw_p;
    TO* const dst = blockIdx.y ? db : dw;
    const int rl = threadIdx.x >> 2, cg = threadIdx.x & 3;
    const int c0 = blockIdx.x * 16 + 4 * cg;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c0 < cols) {      // cols % 4 == 0 (host)
#pragma unroll 8
        for (int r = rl; r < n_part; r += 64) {
            const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)r * cols + c0);
            a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w;
        }
    }
    red[rl][4 * cg + 0] = a0; red[rl][4 * cg + 1] = a1; red[rl][4 * cg + 2] = a2; red[rl][4 * cg + 3] = a3;
    __syncthreads();
    if (threadIdx.x < 16) {
        float t = 0.f;
#pragma unroll 8
        for (int r = 0; r < 64; ++r) t += red[r][threadIdx.x];
        const int c = blockIdx.x * 16 + threadIdx.x;
        if (c < cols) dst[c] = static_cast<TO>(t);
    }
}
}  // namespace vms

extern "C" int vms_layer_norm_bwd_finish(const float* dw_partial, const float* db_partial, int n_partials, int cols, void* dw, void* db,
                                         int out_dtype, void* stream) {
    VMS_CHECK(dw_partial && dw && n_partials > 0 && cols > 0, "dw_partial, dw, n_partials, cols are required");
    VMS_CHECK((db_partial == nullptr) == (db == nullptr), "db comes with db_partial");
    VMS_CHECK(cols % 4 == 0 && aligned16(dw_partial) && (!db_partial || aligned16(db_partial)), "cols % 4 == 0 and 16-byte aligned partial rows");
    VMS_CHECK(out_dtype == VMS_F32 || out_dtype == VMS_F16 || out_dtype == VMS_BF16, "out dtype must be fp32/fp16/bf16");
    const dim3 grid((cols + 15) / 16, db_partial ? 2 : 1), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (out_dtype == VMS_F32) hipLaunchKernelGGL((vms::norm_bwd_finish_kernel<float>), grid, block, 0, s, dw_partial, db_partial, n_partials, cols, static_cast<float*>(dw), static_cast<float*>(db));
    else if (out_dtype == VMS_F16) hipLaunchKernelGGL((vms::norm_bwd_finish_kernel<vms::f16_t>), grid, block, 0, s, dw_partial, db_partial, n_partials, cols, static_cast<vms::f16_t*>(dw), static_cast<vms::f16_t*>(db));
    else hipLaunchKernelGGL((vms::norm_bwd_finish_kernel<vms::bf16_t>), grid, block, 0, s, dw_partial, db_partial, n_partials, cols, static_cast<vms::bf16_t*>(dw), static_cast<vms::bf16_t*>(db));
    VMS_LAUNCH_CHECK();
    return VMS_OK;
}

extern "C" int vms_sizeof_norm_params(void) { return (int)sizeof(vms_norm_params); }
extern "C" int vms_sizeof_norm_bwd_params(void) { return (int)sizeof(vms_norm_bwd_params); }
