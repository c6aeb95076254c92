// scan_bwd_helpers.h -- what the whole-vector backward scan kernels share (selective_scan_bwd_pair.hip; round 5's 128-VGPR experiment, profiles/r05_bwd_occ4.patch, shares them):
// the lane's 8 elements as raw 16-byte vectors, direction-aware stores, DPP row scans, packed-fma and LDS-barrier wrappers.
#pragma once
#include "vms_common.h"

#ifndef VMS_BWD_ST_NT
#define VMS_BWD_ST_NT 1     /* 0 (A/B builds): du / ddelta leave as ordinary stores */
#endif
#ifndef VMS_BWD_LOAD_NT
#define VMS_BWD_LOAD_NT 1   /* 0 (A/B builds): the backward's row data as ordinary loads */
#endif
namespace vms {

constexpr int kBN = 16;   // dstate
constexpr int kBK = 8;    // elements per lane
constexpr int kBQ = 8;    // row quads (waves) per workgroup
constexpr int kBRows = 4 * kBQ;
constexpr int kBSG = 2;            // states between two row reductions
constexpr int kCH = 16 * kBK;      // elements per row per iteration (128)
constexpr int kBcFloats = 2 * kBN * kCH;              // [tensor][state][position]
constexpr int kSlabFloats = kBSG * 2 * kBRows * kCH;  // one pair of states: [state % 2][tensor][row][position]

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) f2 lds_f2;
typedef __attribute__((address_space(3))) f32x4 lds_f4;

template <typename T, bool REV>
struct RawB {
    static constexpr int EPV = 16 / sizeof(T);
    vec_t<T, EPV> v[kBK / EPV];
    // seqlen % K == 0: a lane's elements are all valid or all past the end; invalid lanes read the (always
    // valid) start of the buffer and are neutralised by the caller (delta = 0 / masked stores)
    __device__ __forceinline__ void load(const T* __restrict__ base, uint32_t off, bool valid) {
        const vec_t<T, EPV>* vp = reinterpret_cast<const vec_t<T, EPV>*>(base + (valid ? off : 0u));
#pragma unroll
        for (int i = 0; i < kBK / EPV; ++i) v[i] = vp[i];
    }
    // the same for data this launch touches once (u, delta, dout, z, out): `nt` loads leave the L2 lines of the dB / dC
    // atomics, which 32 workgroups per batch revisit, in place
    __device__ __forceinline__ void load_stream(const T* __restrict__ base, uint32_t off, bool valid) {
        const vec_t<T, EPV>* vp = reinterpret_cast<const vec_t<T, EPV>*>(base + (valid ? off : 0u));
#pragma unroll
        for (int i = 0; i < kBK / EPV; ++i) {
            v[i] = VMS_BWD_LOAD_NT ? __builtin_nontemporal_load(&vp[i]) : vp[i];
        }
    }
    // signed offset: a partly valid vector of a padded B / C row may start before the row (REV) -- vms_hip.h bc_pad
    // (`safe`: the offset a lane beyond the row reads instead -- 0 left-to-right, seqlen - K right-to-left: see RawP::load_s)
    __device__ __forceinline__ void load_s(const T* __restrict__ base, int32_t off, bool valid, int32_t safe) {
        const vec_t<T, EPV>* vp = reinterpret_cast<const vec_t<T, EPV>*>(base + (valid ? off : safe));
#pragma unroll
        for (int i = 0; i < kBK / EPV; ++i) v[i] = vp[i];
    }
    // ragged rows: the nv (< K) valid logical elements [l0, l0 + nv) of the row one by one, zeros behind them
    __device__ __forceinline__ void load_partial(const T* __restrict__ row, int l0, int L, int nv) {
#pragma unroll
        for (int i = 0; i < kBK; ++i) {
            const int e = REV ? kBK - 1 - i : i;
            v[e / EPV][e % EPV] = i < nv ? row[REV ? L - 1 - (l0 + i) : l0 + i] : static_cast<T>(0.f);
        }
    }
    __device__ __forceinline__ float at(int i) const {
        const int e = REV ? kBK - 1 - i : i;
        return static_cast<float>(v[e / EPV][e % EPV]);
    }
};
template <typename T, bool REV>
__device__ __forceinline__ void store_partial_b(T* __restrict__ row, int l0, int L, int nv, const float (&in)[kBK]) {
#pragma unroll
    for (int i = 0; i < kBK; ++i)
        if (i < nv) row[REV ? L - 1 - (l0 + i) : l0 + i] = static_cast<T>(in[i]);
}
template <typename T, bool REV>
__device__ __forceinline__ void store_b(T* __restrict__ ptr, const float (&in)[kBK]) {
    constexpr int EPV = 16 / sizeof(T);
    using V = vec_t<T, EPV>;
#pragma unroll
    for (int v = 0; v < kBK / EPV; ++v) {
        V t;
#pragma unroll
        for (int e = 0; e < EPV; ++e) t[e] = static_cast<T>(in[REV ? kBK - 1 - (v * EPV + e) : v * EPV + e]);
        reinterpret_cast<V*>(ptr)[v] = t;
    }
}
template <typename T, bool REV>
__device__ __forceinline__ void store_stream_b(T* __restrict__ ptr, const float (&in)[kBK]) {
    constexpr int EPV = 16 / sizeof(T);
    using V = vec_t<T, EPV>;
#pragma unroll
    for (int v = 0; v < kBK / EPV; ++v) {
        V t;
#pragma unroll
        for (int e = 0; e < EPV; ++e) t[e] = static_cast<T>(in[REV ? kBK - 1 - (v * EPV + e) : v * EPV + e]);
        if (VMS_BWD_ST_NT) __builtin_nontemporal_store(t, reinterpret_cast<V*>(ptr) + v);
        else reinterpret_cast<V*>(ptr)[v] = t;
    }
}

// element offset (inside a row's x) of the state after the first 128 (e128 + 1) elements for state n: the 128-element
// sub-checkpoints of x_has_sub == 1, or every 16th of the 8-element checkpoints of x_has_sub == 3 (vms_hip.h)
__device__ __forceinline__ uint32_t x_sub_off(int e128, int n, int pitch, bool lane_ckpt, int dstate) {
    return lane_ckpt ? (uint32_t)((e128 >> 4) * pitch + 2 * dstate + ((n >> 2) * 256 + (e128 & 15) * 16 + 15) * 4 + (n & 3))
                     : (uint32_t)((e128 >> 4) * pitch + 2 * dstate + (e128 & 15) * dstate + n);
}

template <int CTRL>
__device__ __forceinline__ float bdpp(float old, float src) {
    return dpp_mov<CTRL, 0xf>(old, src);
}
constexpr int DPP_B_ROW_ROR = 0x120;

// Forward inclusive scan of (pa, px) and suffix inclusive scan of (ra, rg) inside each 16-lane row,
// interleaved: one DPP-fused VOP2 per monoid component and step (x += dpp(x) * a ; a *= dpp(a); lanes
// whose DPP source falls outside the row are not written = identity).  The interleaving also provides the
// 2 wait states a DPP read needs after a VALU write of its source.
__device__ __forceinline__ void row_scan_pair_b(float& pa, float& px, float& ra, float& rg) {
#define VMS_STEP(S)                                                                   \
    "v_fmac_f32_dpp %0, %0, %1 row_shr:" #S " row_mask:0xf bank_mask:0xf\n\t"          \
    "v_fmac_f32_dpp %2, %2, %3 row_shl:" #S " row_mask:0xf bank_mask:0xf\n\t"          \
    "v_mul_f32_dpp %1, %1, %1 row_shr:" #S " row_mask:0xf bank_mask:0xf\n\t"           \
    "v_mul_f32_dpp %3, %3, %3 row_shl:" #S " row_mask:0xf bank_mask:0xf\n\t"
    asm volatile("s_nop 1\n\t" VMS_STEP(1) VMS_STEP(2) VMS_STEP(4) VMS_STEP(8) "s_nop 1"
                 : "+v"(px), "+v"(pa), "+v"(rg), "+v"(ra));
#undef VMS_STEP
}

__device__ __forceinline__ void row_scan_suffix_b(float& ra, float& rg) {
#define VMS_STEP(S)                                                                   \
    "v_fmac_f32_dpp %0, %0, %1 row_shl:" #S " row_mask:0xf bank_mask:0xf\n\t"          \
    "v_mul_f32_dpp %1, %1, %1 row_shl:" #S " row_mask:0xf bank_mask:0xf\n\t"           \
    "s_nop 0\n\t"
    asm volatile("s_nop 1\n\t" VMS_STEP(1) VMS_STEP(2) VMS_STEP(4) VMS_STEP(8) "s_nop 0"
                 : "+v"(rg), "+v"(ra));
#undef VMS_STEP
}

// one v_pk_fma_f32 (the backend splits a <2 x float> fma whose operands were assembled from scalars)
__device__ __forceinline__ f2 pk_fma_b(f2 a, f2 b, f2 c) {
    f2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// the same where an operand comes straight from v_exp_f32: gfx950 wants one wait state between a transcendental's write and a
// non-transcendental VALU read of it, and the hazard recognizer does not look inside an asm statement (without the s_nop the
// carry pass read stale exponentials wherever the scheduler put the fma directly behind the exp)
__device__ __forceinline__ f2 pk_fma_after_trans_b(f2 a, f2 b, f2 c) {
    f2 r;
    asm("s_nop 0\n\tv_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// workgroup barrier that orders LDS traffic only (no vmcnt drain)
__device__ __forceinline__ void lds_barrier_b() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ float row_allsum_b(float v) {
    v += bdpp<DPP_B_ROW_ROR + 1>(0.f, v);
    v += bdpp<DPP_B_ROW_ROR + 2>(0.f, v);
    v += bdpp<DPP_B_ROW_ROR + 4>(0.f, v);
    v += bdpp<DPP_B_ROW_ROR + 8>(0.f, v);
    return v;
}


}  // namespace vms
