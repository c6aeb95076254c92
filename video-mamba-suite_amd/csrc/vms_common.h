// vms_common.h -- shared device helpers for the gfx950 (CDNA4, wave64) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#include "../../include/vms_hip.h"

namespace vms {

constexpr int kWave = 64;
constexpr float kLog2e = 1.4426950408889634f;

using bf16_t = __bf16;
using f16_t = _Float16;

// explicit LDS (address space 3) pointer type: keeps ds_* instructions (a generic float* that
// points into __shared__ memory would be accessed with flat_* and, for atomics, a run-time
// aperture check)
typedef __attribute__((address_space(3))) float lds_f32;
__device__ __forceinline__ void lds_atomic_add(lds_f32* p, float v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <typename T, int N> struct VecT { using type = __attribute__((ext_vector_type(N))) T; };
template <typename T, int N> using vec_t = typename VecT<T, N>::type;

template <typename T> struct DTypeOf;
template <> struct DTypeOf<float> { static constexpr int v = VMS_F32; };
template <> struct DTypeOf<f16_t> { static constexpr int v = VMS_F16; };
template <> struct DTypeOf<bf16_t> { static constexpr int v = VMS_BF16; };

// ---- elementwise math (one v_exp/v_log/v_rcp each) -----------------------------------------
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * kLog2e); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
__device__ __forceinline__ float sigmoidf_(float x) { return fast_rcp(1.f + fast_exp(-x)); }

// softplus with the reference's threshold (selective_scan_fwd_kernel.cuh:153-156); log1p(t) is
// evaluated as log(w) + (t - (w - 1)) / w, w = 1 + t: the first-order correction for the rounding of
// 1 + t keeps full relative accuracy for small t (w == 1 gives t itself) with one rcp and no select.
__device__ __forceinline__ float softplusf_(float x) {
    float t = fast_exp(x);
    float w = 1.f + t;
    float r = fmaf(t - (w - 1.f), fast_rcp(w), fast_log(w));
    return x <= 20.f ? r : x;
}

// ---- blocked loads / stores: K consecutive elements per lane, widened to float ---------------
// VEC: 16-byte vector accesses (caller guarantees 16 B alignment of ptr and that all K
// elements are in range); otherwise element-wise with the bound `n_valid` (elements
// available from ptr), padding with `pad`.
template <typename T, int K, bool VEC>
__device__ __forceinline__ void load_blocked(const T* __restrict__ ptr, int n_valid, float (&out)[K],
                                             float pad = 0.f) {
    constexpr int EPV = 16 / sizeof(T);  // elements per 16-byte vector
    if (VEC && n_valid >= K) {
        static_assert(K % EPV == 0, "K must be a multiple of the 16B vector width");
        using V = vec_t<T, EPV>;
        const V* vp = reinterpret_cast<const V*>(ptr);
#pragma unroll
        for (int v = 0; v < K / EPV; ++v) {
            V t = vp[v];
#pragma unroll
            for (int e = 0; e < EPV; ++e) out[v * EPV + e] = static_cast<float>(t[e]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < K; ++i) out[i] = i < n_valid ? static_cast<float>(ptr[i]) : pad;
    }
}

template <typename T, int K, bool VEC>
__device__ __forceinline__ void store_blocked(T* __restrict__ ptr, int n_valid, const float (&in)[K]) {
    constexpr int EPV = 16 / sizeof(T);
    if (VEC && n_valid >= K) {
        using V = vec_t<T, EPV>;
        V* vp = reinterpret_cast<V*>(ptr);
#pragma unroll
        for (int v = 0; v < K / EPV; ++v) {
            V t;
#pragma unroll
            for (int e = 0; e < EPV; ++e) t[e] = static_cast<T>(in[v * EPV + e]);
            vp[v] = t;
        }
    } else {
#pragma unroll
        for (int i = 0; i < K; ++i)
            if (i < n_valid) ptr[i] = static_cast<T>(in[i]);
    }
}

// Direction-aware variants: the K LOGICAL elements [l0, l0+K) of a row of length L, where logical
// position t is physical position (rev ? L-1-t : t).  Used by the kernels' `reverse` mode, which
// reads / writes the sequence right-to-left instead of running on flipped copies.
template <typename T, int K, bool VEC>
__device__ __forceinline__ void load_dir(const T* __restrict__ row, int l0, int L, bool rev, float (&out)[K],
                                         float pad = 0.f) {
    constexpr int EPV = 16 / sizeof(T);
    const int nv = L - l0;
    if (!rev) {
        load_blocked<T, K, VEC>(row + l0, nv, out, pad);
    } else if (VEC && nv >= K && (L % EPV) == 0) {
        using V = vec_t<T, EPV>;
        const V* vp = reinterpret_cast<const V*>(row + (L - l0 - K));
#pragma unroll
        for (int v = 0; v < K / EPV; ++v) {
            V t = vp[v];
#pragma unroll
            for (int e = 0; e < EPV; ++e) out[K - 1 - (v * EPV + e)] = static_cast<float>(t[e]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < K; ++i) out[i] = i < nv ? static_cast<float>(row[L - 1 - l0 - i]) : pad;
    }
}

template <typename T, int K, bool VEC>
__device__ __forceinline__ void store_dir(T* __restrict__ row, int l0, int L, bool rev, const float (&in)[K]) {
    constexpr int EPV = 16 / sizeof(T);
    const int nv = L - l0;
    if (!rev) {
        store_blocked<T, K, VEC>(row + l0, nv, in);
    } else if (VEC && nv >= K && (L % EPV) == 0) {
        using V = vec_t<T, EPV>;
        V* vp = reinterpret_cast<V*>(row + (L - l0 - K));
#pragma unroll
        for (int v = 0; v < K / EPV; ++v) {
            V t;
#pragma unroll
            for (int e = 0; e < EPV; ++e) t[e] = static_cast<T>(in[K - 1 - (v * EPV + e)]);
            vp[v] = t;
        }
    } else {
#pragma unroll
        for (int i = 0; i < K; ++i)
            if (i < nv) row[L - 1 - l0 - i] = static_cast<T>(in[i]);
    }
}

// ---- DPP cross-lane primitives (wave64 = 4 rows of 16 lanes) ---------------------------------
// dpp_mov<CTRL,ROWMASK>(old, src): lane reads `src` of the lane selected by CTRL; lanes whose
// source does not exist or whose row is masked off keep `old`.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float old, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old),
                                                                  __builtin_bit_cast(int, src), CTRL,
                                                                  ROW_MASK, 0xf, false));
}
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
constexpr int DPP_ROW_SHL1 = 0x101, DPP_ROW_SHL2 = 0x102, DPP_ROW_SHL4 = 0x104, DPP_ROW_SHL8 = 0x108;
constexpr int DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;
constexpr int DPP_WAVE_SHR1 = 0x138, DPP_WAVE_SHL1 = 0x130;

// The scan monoid of the recurrence x <- a*x + b (reference: SSMScanOp, selective_scan_common.h:
// 110-115): (a0,b0) then (a1,b1) = (a1*a0, a1*b0 + b1).
// Inclusive scan over the 64 lanes in lane order (lane 0 = earliest).
__device__ __forceinline__ void wave_scan_inclusive(float& a, float& x) {
#define VMS_SCAN_STEP(C, M)                 \
    {                                       \
        float xp = dpp_mov<C, M>(0.f, x);   \
        float ap = dpp_mov<C, M>(1.f, a);   \
        x = fmaf(a, xp, x);                 \
        a = a * ap;                         \
    }
    VMS_SCAN_STEP(DPP_ROW_SHR1, 0xf)
    VMS_SCAN_STEP(DPP_ROW_SHR2, 0xf)
    VMS_SCAN_STEP(DPP_ROW_SHR4, 0xf)
    VMS_SCAN_STEP(DPP_ROW_SHR8, 0xf)
    VMS_SCAN_STEP(DPP_ROW_BCAST15, 0xa)
    VMS_SCAN_STEP(DPP_ROW_BCAST31, 0xc)
#undef VMS_SCAN_STEP
}

// Same monoid scanned from the LAST lane towards lane 0 (suffix scan): after the call lane i
// holds the composition of lanes i..63 applied in the order 63, 62, ..., i, i.e. for the
// adjoint recurrence g <- a*g + c running from high l to low l.
__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ void wave_scan_inclusive_reverse(float& a, float& x) {
#define VMS_RSCAN_STEP(C)                   \
    {                                       \
        float xp = dpp_mov<C, 0xf>(0.f, x); \
        float ap = dpp_mov<C, 0xf>(1.f, a); \
        x = fmaf(a, xp, x);                 \
        a = a * ap;                         \
    }
    VMS_RSCAN_STEP(DPP_ROW_SHL1)
    VMS_RSCAN_STEP(DPP_ROW_SHL2)
    VMS_RSCAN_STEP(DPP_ROW_SHL4)
    VMS_RSCAN_STEP(DPP_ROW_SHL8)
#undef VMS_RSCAN_STEP
    // cross-row: suffix totals of the rows to the right sit in lanes 16, 32, 48
    const float a1 = readlane_f(a, 16), x1 = readlane_f(x, 16);
    const float a2 = readlane_f(a, 32), x2 = readlane_f(x, 32);
    const float a3 = readlane_f(a, 48), x3 = readlane_f(x, 48);
    // suffix entering row 2 = row3 ; row 1 = row2 after row3 ; row 0 = row1 after (row2 after row3)
    const float a23 = a2 * a3, x23 = fmaf(a2, x3, x2);
    const float a123 = a1 * a23, x123 = fmaf(a1, x23, x1);
    const int row = (threadIdx.x & 63) >> 4;
    const float sa = row == 0 ? a123 : (row == 1 ? a23 : (row == 2 ? a3 : 1.f));
    const float sx = row == 0 ? x123 : (row == 1 ? x23 : (row == 2 ? x3 : 0.f));
    x = fmaf(a, sx, x);
    a = a * sa;
}

__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<DPP_ROW_SHR1, 0xf>(0.f, v);
    v += dpp_mov<DPP_ROW_SHR2, 0xf>(0.f, v);
    v += dpp_mov<DPP_ROW_SHR4, 0xf>(0.f, v);
    v += dpp_mov<DPP_ROW_SHR8, 0xf>(0.f, v);
    v += dpp_mov<DPP_ROW_BCAST15, 0xa>(0.f, v);
    v += dpp_mov<DPP_ROW_BCAST31, 0xc>(0.f, v);
    return readlane_f(v, 63);  // total, wave-uniform
}

// ---- host-side plumbing (vms_host.hip) ---------------------------------------------------------------
void set_error(const char* fmt, ...);      // thread-local message behind vms_last_error()
const char* last_error();
void set_last_kernel(const char* name);    // thread-local name behind vms_last_kernel()
// compute units of the calling thread's current device; immutable per device, looked up once per device
int device_cu_count();
// Runs fn() once per device (the calling thread's current device), from whichever thread gets there first; every
// caller waits for it and gets its hipError_t.  For hipFuncSetAttribute(MaxDynamicSharedMemorySize), which a
// kernel needs before its first launch with more than 64 KB of LDS on EACH device it runs on.
constexpr int kMaxDevices = 64;
struct PerDeviceOnce {
    std::once_flag flag[kMaxDevices];
    hipError_t rc[kMaxDevices];
    template <typename F>
    hipError_t run(F&& fn) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return fn();  // uncached, still correct
        std::call_once(flag[dev], [&] { rc[dev] = fn(); });
        return rc[dev];
    }
};
// reverse_from (vms_hip.h, ABI v5): the problem as its left-to-right and right-to-left sub-batches (selective_scan_fwd.hip)
void scan_fwd_sub_batches(const vms_scan_fwd_params& p, vms_scan_fwd_params& lo, vms_scan_fwd_params& hi);
bool scan_fwd_pair_native_mixed(const vms_scan_fwd_params& p);   // selective_scan_fwd_pair.hip: reverse_from in one launch
bool scan_bwd_pair_native_mixed(const vms_scan_bwd_params& q);   // selective_scan_bwd_pair.hip
// scan kernel generation a call may use (vms_hip.h vms_scan_impl): AUTO = PAIR
inline bool scan_impl_valid(int impl) { return impl == VMS_IMPL_AUTO || impl == VMS_IMPL_GENERIC || impl == VMS_IMPL_PAIR; }
inline int scan_impl_level(const vms_scan_fwd_params& p) { return p.impl == VMS_IMPL_AUTO ? VMS_IMPL_PAIR : p.impl; }

#define VMS_CHECK(cond, ...)                                              \
    do {                                                                  \
        if (!(cond)) {                                                    \
            ::vms::set_error("vms check failed: " #cond " -- " __VA_ARGS__); \
            return VMS_ERR_INVALID_ARG;                                   \
        }                                                                 \
    } while (0)

#define VMS_LAUNCH_CHECK()                                                        \
    do {                                                                          \
        hipError_t e_ = hipGetLastError();                                        \
        if (e_ != hipSuccess) {                                                   \
            ::vms::set_error("kernel launch failed: %s", hipGetErrorString(e_)); \
            return VMS_ERR_LAUNCH;                                                \
        }                                                                         \
    } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline bool mult16(int64_t stride_elems, int elem_size) { return ((stride_elems * elem_size) & 15) == 0; }

}  // namespace vms
