// param_prep.hip -- the per-step preparation of a block's parameters as ONE launch (vms_hip.h vms_param_prep).
//
// Under autocast every step of the reference's block casts its projection weights to the compute dtype and evaluates
// A = -exp(A_log) (mamba_simple.py:230, 246; selective_scan_interface.py:169-171: one small kernel each); this build also
// keeps in_proj's weight as a K-contiguous (d_model, channels) copy (ops/projections.py).  In the step trace of the
// (8, 8192, 1024) block those were six launches = 61 us between the large kernels (profiles/r03z2_step_trace.txt: a 16.7 us
// strided transpose-copy, 20 us of multi-tensor casts, 12.6 us for the two A, 2 x 5.7 us for out_proj's weight).  Here: up to 8
// jobs -- cast, cast + transpose, -exp -- as one grid; a workgroup owns one 64 x 64 tile of one job; transposes go through
// a padded LDS tile so that both sides move whole 128 / 256-byte runs.
#include "vms_common.h"

namespace vms {

namespace {

constexpr int kTile = 64;

template <typename T>
__device__ __forceinline__ float ld_as_float(const void* p, int64_t i) { return static_cast<float>(static_cast<const T*>(p)[i]); }

__device__ __forceinline__ float ld_f(const void* p, int64_t i, int dt) {
    return dt == VMS_F32 ? ld_as_float<float>(p, i) : dt == VMS_F16 ? ld_as_float<f16_t>(p, i) : ld_as_float<bf16_t>(p, i);
}
__device__ __forceinline__ void st_f(void* p, int64_t i, int dt, float v) {
    if (dt == VMS_F32) static_cast<float*>(p)[i] = v;
    else if (dt == VMS_F16) static_cast<f16_t*>(p)[i] = static_cast<f16_t>(v);
    else static_cast<bf16_t*>(p)[i] = static_cast<bf16_t>(v);
}

__global__ __launch_bounds__(256) void param_prep_kernel(const vms_prep_params p) {
    __shared__ float tile[kTile][kTile + 1];
    // which job, which tile of it (n_jobs <= 8: a scan over kernel arguments)
    int j = 0, t = blockIdx.x;
    for (; j < p.n_jobs; ++j) {
        const int nt = ((p.job[j].rows + kTile - 1) / kTile) * ((p.job[j].cols + kTile - 1) / kTile);
        if (t < nt) break;
        t -= nt;
    }
    if (j >= p.n_jobs) return;
    const vms_prep_job& q = p.job[j];
    const int tc = (q.cols + kTile - 1) / kTile;
    const int r0 = (t / tc) * kTile, c0 = (t % tc) * kTile;
    const int64_t dcs = q.dst_col_stride > 0 ? q.dst_col_stride : 1;   // destination columns `dcs` elements apart (interleaved halves)
    // whole tiles of an fp32 matrix going to a 16-bit one (every job of a ViM block but the two A): 16 consecutive values
    // per thread, 16-byte accesses on both sides
    if (q.op != VMS_PREP_NEG_EXP && dcs == 1 && q.src_dtype == VMS_F32 && q.dst_dtype != VMS_F32 && r0 + kTile <= q.rows && c0 + kTile <= q.cols &&
        (q.src_row_stride & 3) == 0 && (q.dst_row_stride & 7) == 0 && ((uintptr_t)q.src & 15) == 0 && ((uintptr_t)q.dst & 15) == 0) {
        const int row = threadIdx.x >> 2, seg = (threadIdx.x & 3) * 16;
        const float4* sp = reinterpret_cast<const float4*>(static_cast<const float*>(q.src) + (int64_t)(r0 + row) * q.src_row_stride + c0 + seg);
        float v[16];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 f = sp[k];
            v[4 * k] = f.x; v[4 * k + 1] = f.y; v[4 * k + 2] = f.z; v[4 * k + 3] = f.w;
        }
        int64_t o = (int64_t)(r0 + row) * q.dst_row_stride + c0 + seg;
        if (q.op == VMS_PREP_CAST_T) {
#pragma unroll
            for (int k = 0; k < 16; ++k) tile[row][seg + k] = v[k];
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = tile[seg + k][row];   // dst row c0 + row, dst columns r0 + seg ..
            o = (int64_t)(c0 + row) * q.dst_row_stride + r0 + seg;
        }
        using V8 = __attribute__((ext_vector_type(8))) unsigned short;
        V8 lo, hi;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (q.dst_dtype == VMS_BF16) {
                lo[k] = __builtin_bit_cast(unsigned short, static_cast<bf16_t>(v[k]));
                hi[k] = __builtin_bit_cast(unsigned short, static_cast<bf16_t>(v[8 + k]));
            } else {
                lo[k] = __builtin_bit_cast(unsigned short, static_cast<f16_t>(v[k]));
                hi[k] = __builtin_bit_cast(unsigned short, static_cast<f16_t>(v[8 + k]));
            }
        }
        V8* dp = reinterpret_cast<V8*>(static_cast<unsigned short*>(q.dst) + o);
        dp[0] = lo;
        dp[1] = hi;
        return;
    }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 columns x 4 row groups
    if (q.op != VMS_PREP_CAST_T) {
        // all 16 loads first (clamped addresses, no branch around them): one round trip instead of 16 -- the -exp(A_log) tiles,
        // 16 columns wide, used to set this kernel's duration (12 us for a 32 KB job; round 4)
        float v[kTile / 4];
#pragma unroll
        for (int k = 0; k < kTile / 4; ++k) {
            const int r = r0 + ty + 4 * k, c = c0 + tx;
            const bool ok = r < q.rows && c < q.cols;
            v[k] = ld_f(q.src, ok ? (int64_t)r * q.src_row_stride + c : 0, q.src_dtype);
        }
#pragma unroll
        for (int k = 0; k < kTile / 4; ++k) {
            const int r = r0 + ty + 4 * k, c = c0 + tx;
            if (r < q.rows && c < q.cols)
                st_f(q.dst, (int64_t)r * q.dst_row_stride + c * dcs, q.dst_dtype, q.op == VMS_PREP_NEG_EXP ? -expf(v[k]) : v[k]);
        }
        return;
    }
    // dst (cols, rows) <- src (rows, cols)
    {
        float v[kTile / 4];
#pragma unroll
        for (int k = 0; k < kTile / 4; ++k) {
            const int r = r0 + ty + 4 * k, c = c0 + tx;
            const bool ok = r < q.rows && c < q.cols;
            const float f = ld_f(q.src, ok ? (int64_t)r * q.src_row_stride + c : 0, q.src_dtype);
            v[k] = ok ? f : 0.f;
        }
#pragma unroll
        for (int k = 0; k < kTile / 4; ++k) tile[ty + 4 * k][tx] = v[k];
    }
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < kTile / 4; ++k) {
        const int c = c0 + ty + 4 * k, r = r0 + tx;   // dst row c, dst column r
        if (r < q.rows && c < q.cols) st_f(q.dst, (int64_t)c * q.dst_row_stride + r * dcs, q.dst_dtype, tile[tx][ty + 4 * k]);
    }
}

}  // namespace
}  // namespace vms

using namespace vms;

extern "C" int vms_param_prep(const vms_prep_params* pp, void* stream) {
    VMS_CHECK(pp != nullptr, "null params");
    const vms_prep_params& p = *pp;
    VMS_CHECK(p.n_jobs >= 0 && p.n_jobs <= VMS_PREP_MAX_JOBS, "n_jobs out of range");
    int64_t tiles = 0;
    for (int j = 0; j < p.n_jobs; ++j) {
        const vms_prep_job& q = p.job[j];
        VMS_CHECK(q.src && q.dst && q.rows > 0 && q.cols > 0, "job: src, dst and a non-empty shape are required");
        VMS_CHECK(q.op >= VMS_PREP_CAST && q.op <= VMS_PREP_NEG_EXP, "job: unknown op");
        VMS_CHECK(q.src_dtype >= VMS_F32 && q.src_dtype <= VMS_BF16 && q.dst_dtype >= VMS_F32 && q.dst_dtype <= VMS_BF16, "job: dtype");
        VMS_CHECK(q.dst_col_stride >= 0 && q.src_row_stride >= q.cols &&
                      q.dst_row_stride >= (int64_t)(q.op == VMS_PREP_CAST_T ? q.rows : q.cols) * (q.dst_col_stride > 0 ? q.dst_col_stride : 1),
                  "job: row strides");
        tiles += (int64_t)((q.rows + kTile - 1) / kTile) * ((q.cols + kTile - 1) / kTile);
    }
    if (tiles == 0) return VMS_OK;
    VMS_CHECK(tiles < (1 << 30), "too many tiles");
    set_last_kernel("param_prep");
    hipLaunchKernelGGL(param_prep_kernel, dim3((unsigned)tiles), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    VMS_LAUNCH_CHECK();
    return VMS_OK;
}
extern "C" int vms_sizeof_prep_params(void) { return (int)sizeof(vms_prep_params); }

// ---- sum of the K slices of a weight-gradient GEMM ---------------------------------------------------------
// The block's large weight gradients run as a batched GEMM over K slices (one workgroup per CU) whose results are then summed
// (mamba_ssm/ops/projections.py): dst[i] = sum_s src[s * slice_stride + i], 16-bit slices, fp32 accumulation, dst in the
// parameter's dtype.  A thread owns 8 consecutive elements (one 16-byte load per slice, all independent).
namespace vms {
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void sum_slices_kernel(const TI* __restrict__ src, const int n_slices, const int64_t n, const int64_t slice_stride,
                                                         TO* __restrict__ dst) {
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i0 >= n) return;      // n % 8 == 0 (host)
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int s = 0; s < n_slices; ++s) {
        const vec_t<TI, 8> v = *reinterpret_cast<const vec_t<TI, 8>*>(src + (int64_t)s * slice_stride + i0);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += static_cast<float>(v[e]);
    }
    vec_t<TO, 8> o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = static_cast<TO>(acc[e]);
    *reinterpret_cast<vec_t<TO, 8>*>(dst + i0) = o;
}
template <typename TI>
static int launch_sum_slices(const void* src, int n_slices, int64_t n, int64_t slice_stride, void* dst, int dst_dtype, hipStream_t s) {
    const dim3 grid((unsigned)((n / 8 + 255) / 256)), block(256);
    const TI* sp = static_cast<const TI*>(src);
    if (dst_dtype == VMS_F32) hipLaunchKernelGGL((sum_slices_kernel<TI, float>), grid, block, 0, s, sp, n_slices, n, slice_stride, static_cast<float*>(dst));
    else if (dst_dtype == VMS_F16) hipLaunchKernelGGL((sum_slices_kernel<TI, f16_t>), grid, block, 0, s, sp, n_slices, n, slice_stride, static_cast<f16_t*>(dst));
    else hipLaunchKernelGGL((sum_slices_kernel<TI, bf16_t>), grid, block, 0, s, sp, n_slices, n, slice_stride, static_cast<bf16_t*>(dst));
    VMS_LAUNCH_CHECK();
    return VMS_OK;
}
}  // namespace vms

extern "C" int vms_sum_slices(const void* src, int src_dtype, int n_slices, int64_t n, int64_t slice_stride, void* dst, int dst_dtype, void* stream) {
    VMS_CHECK(src && dst && n_slices > 0 && n > 0, "src, dst, n_slices, n are required");
    VMS_CHECK(src_dtype == VMS_BF16 || src_dtype == VMS_F16, "sum_slices: 16-bit slices (bf16 / fp16)");
    VMS_CHECK(dst_dtype == VMS_F32 || dst_dtype == VMS_F16 || dst_dtype == VMS_BF16, "dst dtype must be fp32/fp16/bf16");
    VMS_CHECK(n % 8 == 0 && slice_stride % 8 == 0 && slice_stride >= n && aligned16(src) && aligned16(dst) && n / 8 < ((int64_t)1 << 38),
              "sum_slices: n and slice_stride multiples of 8 elements, 16-byte aligned src / dst");
    hipStream_t s = static_cast<hipStream_t>(stream);
    set_last_kernel("sum_slices");
    return src_dtype == VMS_BF16 ? launch_sum_slices<bf16_t>(src, n_slices, n, slice_stride, dst, dst_dtype, s)
                                 : launch_sum_slices<f16_t>(src, n_slices, n, slice_stride, dst, dst_dtype, s);
}
