// selective_scan_rows_fwd.hip -- "rows" forward selective scan for gfx950 (CDNA4, wave64).
//
// Same contract as the reference's selective_scan_fwd_kernel (mamba/csrc/selective_scan/
// selective_scan_fwd_kernel.cuh:67-303) for the case every model of the suite runs: variable B and C,
// dstate 16, real A.  Not a translation: the reference gives a thread block one (batch, dim) row and
// scans it with cub::BlockScan; here
//   * a lane owns one ROW (b, d) and keeps its 16 recurrences in registers, walking the sequence
//     serially -- no cross-lane scan at all; a wave = 64 consecutive channels of one batch / group, so
//     B[., n, l] and C[., n, l] are wave-uniform and are fed to the VALU as SGPR pairs (scalar loads,
//     software-prefetched one element ahead);
//   * the 16 states are processed as 8 PAIRS with packed fp32 ops (v_pk_mul_f32 / v_pk_fma_f32):
//     per (element, state pair) = pk_mul, 2 x v_exp_f32, pk_mul, pk_fma, pk_fma.  On gfx950 a packed op
//     issues in 4 cycles for 2 lanes-ops also next to transcendentals, where scalar fp32 ops
//     degrade from ~2.2 to ~3.8 cycles (tools/microbench/microbench5.hip, profiles/r01_microbench_issue.txt);
//   * parallelism along the sequence comes from 128-element chunks: pass 1 computes every chunk's local
//     end state and sum of delta, a tiny carry kernel chains them (and writes the checkpoints the
//     backward pass starts from), pass 2 redoes the recurrence from the true chunk-start state and
//     contracts with C.
// B/C are first rewritten once per call as fp32 [batch][group][position][B0..B15, C0..C15] (scan order),
// so that one s_load_dwordx16 brings all states of one position.
#include "vms_common.h"

namespace vms {

constexpr int kRN = 16;    // dstate
constexpr int kRT = 128;   // chunk length == checkpoint distance
constexpr int kRE = 8;     // elements per inner step
#ifndef VMS_RTE
#define VMS_RTE 16
#endif
constexpr int kRTE = VMS_RTE;   // positions per activation tile
#ifndef VMS_DMA_AUX
#define VMS_DMA_AUX 0
#endif
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));

struct RowsGeom {
    int nch;           // 128-element chunks per row
    int rbpb;          // 64-row blocks per batch
    int rbpg;          // 64-row blocks per group
    int64_t lpad;      // positions per (batch, group) in the bc buffer
    float* bc;         // [batch][group][lpad][32]
    float* agg;        // [batch][rbpb][nch][17][64]: k < 16 local end state, k == 16 sum of delta
    float* hck;        // [batch][rbpb][nch][16][64]: state before chunk c
};

bool scan_rows_eligible(const vms_scan_fwd_params& p) {
    if (!p.is_variable_B || !p.is_variable_C || p.dstate != kRN || p.out_z_accumulate) return false;
    if (p.dim % p.n_groups != 0 || (p.dim / p.n_groups) % 64 != 0) return false;
    if (p.seqlen % kRTE != 0) return false;
    return true;
}

static inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

static RowsGeom rows_geom(const vms_scan_fwd_params& p) {
    RowsGeom g;
    g.nch = (p.seqlen + kRT - 1) / kRT;
    g.rbpb = p.dim / 64;
    g.rbpg = p.dim / p.n_groups / 64;
    g.lpad = (int64_t)g.nch * kRT + kRE;  // + one step: the scalar prefetch runs one position ahead
    g.bc = g.agg = g.hck = nullptr;
    return g;
}
static int64_t rows_bc_bytes(const vms_scan_fwd_params& p, const RowsGeom& g) {
    return align_up((int64_t)p.batch * p.n_groups * g.lpad * 32 * 4, 256);
}
static int64_t rows_agg_bytes(const vms_scan_fwd_params& p, const RowsGeom& g) {
    return align_up((int64_t)p.batch * g.rbpb * g.nch * 17 * 64 * 4, 256);
}
int64_t scan_rows_hck_elems(const vms_scan_fwd_params& p) {
    const RowsGeom g = rows_geom(p);
    return (int64_t)p.batch * g.rbpb * g.nch * 16 * 64;
}
int64_t scan_rows_fwd_ws_bytes(const vms_scan_fwd_params& p) {
    const RowsGeom g = rows_geom(p);
    return rows_bc_bytes(p, g) + rows_agg_bytes(p, g);
}

// scalar loads of one position's 16 B (or C) values; completion is awaited by SWAIT* below, which also
// ties the registers to the instruction stream (the compiler does not track asm loads)
#ifdef VMS_DBG_NOSMEM  // profiling experiment: always the same (cached) line
#define VMS_SLOAD16(dst, base, imm) asm volatile("s_load_dwordx16 %0, %1, 0" : "=&s"(dst) : "s"(g.bc))
#else
#define VMS_SLOAD16(dst, base, imm) asm volatile("s_load_dwordx16 %0, %1, " #imm : "=&s"(dst) : "s"(base))
#endif
#define VMS_SWAIT1(a, dep) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+v"(dep))
#define VMS_SWAIT2(a, b, dep) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b), "+v"(dep))

#define VMS_PAIR(q, p) __builtin_shufflevector(q, q, 2 * (p), 2 * (p) + 1)

// ---- B/C -> fp32, position-major, scan order ------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void scan_rows_prep_kernel(const vms_scan_fwd_params p, const RowsGeom g) {
    const int bg = blockIdx.y;
    const int b = bg / p.n_groups, grp = bg % p.n_groups;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= g.lpad) return;
    float4* dst = reinterpret_cast<float4*>(g.bc + ((int64_t)bg * g.lpad + t) * 32);
    float v[32];
    if (t < p.seqlen) {
        const int64_t l = p.reverse ? p.seqlen - 1 - t : t;
        const T* Bp = static_cast<const T*>(p.B) + (int64_t)b * p.B_batch_stride + (int64_t)grp * p.B_group_stride + l;
        const T* Cp = static_cast<const T*>(p.C) + (int64_t)b * p.C_batch_stride + (int64_t)grp * p.C_group_stride + l;
#pragma unroll
        for (int n = 0; n < kRN; ++n) {
            v[n] = static_cast<float>(Bp[(int64_t)n * p.B_dstate_stride]);
            v[16 + n] = static_cast<float>(Cp[(int64_t)n * p.C_dstate_stride]);
        }
    } else {
#pragma unroll
        for (int n = 0; n < 32; ++n) v[n] = 0.f;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) dst[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

// ---- activation tiles: 64 rows x 16 positions staged through a wave-private LDS image -----------------
// A lane computes on ITS row, but a wave must touch memory in whole row segments (16-byte pieces of 64
// different rows per instruction refetch every 128-byte line 8 times through L1 -- measured 2.5-5x
// slower).  So tiles move with LDS-DMA (global_load_lds_dwordx4: no VGPRs, 1 KiB per instruction, lane l
// fetches segment l % SEG of row l / SEG), land as [64 rows][16 * sizeof(T)] and are read back one row per
// lane; outputs take the same image the other way.  Only the owning wave touches its image: no barriers.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;

template <typename T>
struct Tile {
    static constexpr int EPV = 16 / sizeof(T);          // elements per 16-byte segment
    static constexpr int SEG = kRTE / EPV;              // segments per row (2: 16-bit, 4: fp32)
    static constexpr int RPI = 64 / SEG;                // rows per DMA instruction
    static constexpr int BYTES = 64 * kRTE * sizeof(T);
    // base: element (row 0 of the block, position 0); pt0: first PHYSICAL position of the tile
    static __device__ __forceinline__ void dma_in(const T* base, int64_t d_stride, int pt0, char* img, int lane) {
        const T* g = base + (int64_t)(lane / SEG) * d_stride + pt0 + (lane % SEG) * EPV;
#pragma unroll
        for (int j = 0; j < SEG; ++j)
            __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(g + (int64_t)(j * RPI) * d_stride), (lds_void_t*)(img + j * 1024), 16, 0, VMS_DMA_AUX);
    }
    static __device__ __forceinline__ void flush_out(T* base, int64_t d_stride, int pt0, const char* img, int lane) {
        T* g = base + (int64_t)(lane / SEG) * d_stride + pt0 + (lane % SEG) * EPV;
#pragma unroll
        for (int j = 0; j < SEG; ++j)
            *reinterpret_cast<vec_t<T, EPV>*>(g + (int64_t)(j * RPI) * d_stride) =
                *reinterpret_cast<const vec_t<T, EPV>*>(img + j * 1024 + lane * 16);
    }
};

// the lane's 16 positions of one tile, in SCAN order through at() (REV: the image is right-to-left)
template <typename T, bool REV>
struct Raw16 {
    static constexpr int EPV = 16 / sizeof(T);
    vec_t<T, EPV> v[kRTE / EPV];
    __device__ __forceinline__ void read(const char* img, int lane) {
        const vec_t<T, EPV>* src = reinterpret_cast<const vec_t<T, EPV>*>(img + lane * (kRTE * sizeof(T)));
#pragma unroll
        for (int k = 0; k < kRTE / EPV; ++k) v[k] = src[k];
    }
    __device__ __forceinline__ float at(int i) const {
        const int e = REV ? kRTE - 1 - i : i;
        return static_cast<float>(v[e / EPV][e % EPV]);
    }
};
// 8 results (scan positions h*8 .. h*8+7 of the tile) into the lane's row of an output image
template <typename T, bool REV>
__device__ __forceinline__ void stage8(char* img, int lane, int h, const float (&y)[8]) {
    constexpr int EPV = 16 / sizeof(T);
    const int e0 = REV ? kRTE - 8 - h * 8 : h * 8;  // first physical element of this half
    vec_t<T, EPV>* dst = reinterpret_cast<vec_t<T, EPV>*>(img + lane * (kRTE * sizeof(T)) + e0 * sizeof(T));
#pragma unroll
    for (int k = 0; k < 8 / EPV; ++k) {
        vec_t<T, EPV> o;
#pragma unroll
        for (int e = 0; e < EPV; ++e) o[e] = static_cast<T>(y[REV ? 7 - (k * EPV + e) : k * EPV + e]);
        dst[k] = o;
    }
}
// lanes exchange data through the image: keep the compiler from reordering LDS accesses it can prove
// independent per thread (the LDS itself executes a wave's instructions in order)
#define VMS_LDS_ORDER() asm volatile("" ::: "memory")
#define VMS_WAIT_VM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define VMS_WAIT_LGKM() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// wave -> (batch, 64-row block, chunk): the 4 waves of a workgroup are 4 row blocks of one (batch, chunk),
// chunks are the fastest workgroup index (neighbouring workgroups stream neighbouring pieces of the same rows)
struct RowsWave {
    int b, rb, c, d, grp;
    bool ok;
};
__device__ __forceinline__ RowsWave rows_wave(const vms_scan_fwd_params& p, const RowsGeom& g) {
    RowsWave w;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rbq = (g.rbpb + 3) / 4;
    w.c = blockIdx.x % g.nch;
    const int t = blockIdx.x / g.nch;
    w.b = t / rbq;
    w.rb = (t % rbq) * 4 + wave;
    w.ok = w.rb < g.rbpb;
    w.d = w.rb * 64 + (threadIdx.x & 63);
    w.grp = w.rb / g.rbpg;
    return w;
}

// ---- pass 1: chunk-local end state (from a zero state) and sum of delta ------------------------------
// One position, all 8 state pairs.  Plain vector code: the backend selects v_pk_mul_f32 / v_pk_fma_f32 (the
// splat of delta becomes an op_sel modifier, the B / C pairs stay SGPR operands), schedules around the
// result latencies and inserts only the wait states the hardware needs (hand-placed asm made the hazard
// recognizer pad every dependent pair, and hid the transcendental -> VALU hazard from it).
#define VMS_SPLAT(v, i) __builtin_shufflevector(v[(i) / 2], v[(i) / 2], (i) % 2, (i) % 2)
#define VMS_FOR8(M, ...) M(0, __VA_ARGS__) M(1, __VA_ARGS__) M(2, __VA_ARGS__) M(3, __VA_ARGS__) M(4, __VA_ARGS__) M(5, __VA_ARGS__) M(6, __VA_ARGS__) M(7, __VA_ARGS__)
#define VMS_P1_Q(q, CUR)                                                    \
    {                                                                       \
        const f2 t = ds * A2[q];                                            \
        const f2 a = f2{fast_exp2(t.x), fast_exp2(t.y)};                    \
        x2[q] = __builtin_elementwise_fma(a, x2[q], vs * VMS_PAIR(CUR, q)); \
    }
#define VMS_P1_ELEM(i, CUR, NXT, IMM)                  \
    {                                                  \
        VMS_SWAIT1(CUR, x2[7]);                        \
        VMS_SLOAD16(NXT, bcp, IMM);                    \
        const f2 ds = VMS_SPLAT(d2, i), vs = VMS_SPLAT(v2, i); \
        VMS_FOR8(VMS_P1_Q, CUR)                        \
    }

template <typename T, bool SP, bool REV>
__global__ __launch_bounds__(256) void scan_rows_p1_kernel(const vms_scan_fwd_params p, const RowsGeom g) {
    __shared__ __attribute__((aligned(16))) char smem[4 * 2 * Tile<T>::BYTES];
    const RowsWave w = rows_wave(p, g);
    if (!w.ok) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* const img_u = smem + wave * (2 * Tile<T>::BYTES);
    char* const img_d = img_u + Tile<T>::BYTES;
    const int L = p.seqlen;
    // element (first row of the block, position 0)
    const T* u0 = static_cast<const T*>(p.u) + (int64_t)w.b * p.u_batch_stride + (int64_t)(w.rb * 64) * p.u_d_stride;
    const T* d0 = static_cast<const T*>(p.delta) + (int64_t)w.b * p.delta_batch_stride + (int64_t)(w.rb * 64) * p.delta_d_stride;
    const float bias = p.delta_bias ? static_cast<const float*>(p.delta_bias)[w.d] : 0.f;
    f2 A2[8], x2[8];
    {
        const float* Ap = static_cast<const float*>(p.A) + (int64_t)w.d * p.A_d_stride;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            A2[q] = f2{Ap[(2 * q) * p.A_dstate_stride] * kLog2e, Ap[(2 * q + 1) * p.A_dstate_stride] * kLog2e};
            x2[q] = f2{0.f, 0.f};
        }
    }
    const int t_begin = w.c * kRT;
    const int ntiles = min(kRT, L - t_begin) / kRTE;
    const float* bcp = g.bc + ((int64_t)(w.b * p.n_groups + w.grp) * g.lpad + t_begin) * 32;
    f16v Bq0, Bq1;
    VMS_SLOAD16(Bq0, bcp, 0);
    auto pt = [&](int tile) { const int t0 = t_begin + tile * kRTE; return REV ? L - t0 - kRTE : t0; };
    Tile<T>::dma_in(u0, p.u_d_stride, pt(0), img_u, lane);
    Tile<T>::dma_in(d0, p.delta_d_stride, pt(0), img_d, lane);
    float sd = 0.f;
    for (int tile = 0; tile < ntiles; ++tile) {
        Raw16<T, REV> ru, rd;
        VMS_WAIT_VM();
        ru.read(img_u, lane);
        rd.read(img_d, lane);
        VMS_WAIT_LGKM();
#ifndef VMS_DBG_NOVMEM
        if (tile + 1 < ntiles) {
            Tile<T>::dma_in(u0, p.u_d_stride, pt(tile + 1), img_u, lane);
            Tile<T>::dma_in(d0, p.delta_d_stride, pt(tile + 1), img_d, lane);
        }
#endif
#pragma unroll
        for (int h = 0; h < kRTE / 8; ++h) {
            f2 d2[4], v2[4];
#pragma unroll
            for (int i = 0; i < kRE; ++i) {
                float t = rd.at(h * 8 + i) + bias;
                if (SP) t = softplusf_(t);
                d2[i / 2][i % 2] = t;
                v2[i / 2][i % 2] = t * ru.at(h * 8 + i);
                sd += t;
            }
            VMS_P1_ELEM(0, Bq0, Bq1, 0x80)
            VMS_P1_ELEM(1, Bq1, Bq0, 0x100)
            VMS_P1_ELEM(2, Bq0, Bq1, 0x180)
            VMS_P1_ELEM(3, Bq1, Bq0, 0x200)
            VMS_P1_ELEM(4, Bq0, Bq1, 0x280)
            VMS_P1_ELEM(5, Bq1, Bq0, 0x300)
            VMS_P1_ELEM(6, Bq0, Bq1, 0x380)
            VMS_P1_ELEM(7, Bq1, Bq0, 0x400)
            bcp += kRE * 32;
        }
    }
    VMS_SWAIT1(Bq0, x2[7]);
    float* ag = g.agg + (((int64_t)w.b * g.rbpb + w.rb) * g.nch + w.c) * (17 * 64) + lane;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        ag[(2 * q) * 64] = x2[q].x;
        ag[(2 * q + 1) * 64] = x2[q].y;
    }
    ag[16 * 64] = sd;
}
#undef VMS_P1_ELEM
#undef VMS_P1_Q

// ---- carry: chain the chunk aggregates; writes the chunk-start states and the reference-shaped x -----
__global__ __launch_bounds__(256) void scan_rows_carry_kernel(const vms_scan_fwd_params p, const RowsGeom g) {
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = tid & 63;
    const int n = (tid >> 6) & 15;
    const int64_t brb = tid >> 10;  // b * rbpb + rb
    if (brb >= (int64_t)p.batch * g.rbpb) return;
    const int b = brb / g.rbpb, rb = brb % g.rbpb;
    const int d = rb * 64 + lane;
    const float An = static_cast<const float*>(p.A)[(int64_t)d * p.A_d_stride + (int64_t)n * p.A_dstate_stride] * kLog2e;
    const float* ag = g.agg + brb * g.nch * (17 * 64) + lane;
    float* hk = g.hck + brb * g.nch * (16 * 64) + n * 64 + lane;
    float* xr = static_cast<float*>(p.x) + ((int64_t)b * p.dim + d) * p.n_chunks * (2 * kRN);
    float h = 0.f;
    for (int c = 0; c < g.nch; ++c) {
        hk[(int64_t)c * (16 * 64)] = h;
        const float e = ag[(int64_t)c * (17 * 64) + n * 64];
        const float sd = ag[(int64_t)c * (17 * 64) + 16 * 64];
        h = fmaf(fast_exp2(An * sd), h, e);
        // reference-shaped checkpoints (vms_hip.h): even slot = state after the first 1024 elements of a
        // 2048-chunk, odd slot = state after the chunk (or the sequence)
        const bool last = c == g.nch - 1;
        const int pos = last ? p.seqlen : (c + 1) * kRT;
        if (last || pos % 1024 == 0) {
            const int blk = (pos - 1) / 2048;
            const int r = pos - blk * 2048;
            if (r <= 1024) xr[blk * (2 * kRN) + 2 * n] = h;
            if (r == 2048 || last) xr[blk * (2 * kRN) + 2 * n + 1] = h;
        }
    }
}

// ---- pass 2: the recurrence from the true chunk-start state, contraction with C, gate ------------------
#define VMS_P2_Q(q, i, CB, CC)                                              \
    {                                                                       \
        const f2 t = ds * A2[q];                                            \
        const f2 a = f2{fast_exp2(t.x), fast_exp2(t.y)};                    \
        x2[q] = __builtin_elementwise_fma(a, x2[q], vs * VMS_PAIR(CB, q));  \
        y2[i] = __builtin_elementwise_fma(VMS_PAIR(CC, q), x2[q], y2[i]);   \
    }
#define VMS_P2_ELEM(i, CB, CC, NB, NC, IMMB, IMMC)     \
    {                                                  \
        VMS_SWAIT2(CB, CC, x2[7]);                     \
        VMS_SLOAD16(NB, bcp, IMMB);                    \
        VMS_SLOAD16(NC, bcp, IMMC);                    \
        const f2 ds = VMS_SPLAT(d2, i), vs = VMS_SPLAT(v2, i); \
        VMS_FOR8(VMS_P2_Q, i, CB, CC)                  \
    }

template <typename T, bool HZ, bool SP, bool REV>
__global__ __launch_bounds__(256) void scan_rows_p2_kernel(const vms_scan_fwd_params p, const RowsGeom g) {
    constexpr int NIMG = HZ ? 5 : 3;  // u, delta, out (+ z, out_z)
    __shared__ __attribute__((aligned(16))) char smem[4 * NIMG * Tile<T>::BYTES];
    const RowsWave w = rows_wave(p, g);
    if (!w.ok) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* const img_u = smem + wave * (NIMG * Tile<T>::BYTES);
    char* const img_d = img_u + Tile<T>::BYTES;
    char* const img_o = img_d + Tile<T>::BYTES;
    char* const img_z = img_o + Tile<T>::BYTES;
    char* const img_oz = img_z + Tile<T>::BYTES;
    const int L = p.seqlen;
    const int r0 = w.rb * 64;
    const T* u0 = static_cast<const T*>(p.u) + (int64_t)w.b * p.u_batch_stride + (int64_t)r0 * p.u_d_stride;
    const T* d0 = static_cast<const T*>(p.delta) + (int64_t)w.b * p.delta_batch_stride + (int64_t)r0 * p.delta_d_stride;
    T* o0 = static_cast<T*>(p.out) + (int64_t)w.b * p.out_batch_stride + (int64_t)r0 * p.out_d_stride;
    const T* z0 = HZ ? static_cast<const T*>(p.z) + (int64_t)w.b * p.z_batch_stride + (int64_t)r0 * p.z_d_stride : nullptr;
    T* oz0 = HZ ? static_cast<T*>(p.out_z) + (int64_t)w.b * p.out_z_batch_stride + (int64_t)r0 * p.out_z_d_stride : nullptr;
    const float bias = p.delta_bias ? static_cast<const float*>(p.delta_bias)[w.d] : 0.f;
    const float Dd = p.D ? static_cast<const float*>(p.D)[w.d] : 0.f;
    f2 A2[8], x2[8];
    {
        const float* Ap = static_cast<const float*>(p.A) + (int64_t)w.d * p.A_d_stride;
        const float* hk = g.hck + (((int64_t)w.b * g.rbpb + w.rb) * g.nch + w.c) * (16 * 64) + lane;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            A2[q] = f2{Ap[(2 * q) * p.A_dstate_stride] * kLog2e, Ap[(2 * q + 1) * p.A_dstate_stride] * kLog2e};
            x2[q] = f2{hk[(2 * q) * 64], hk[(2 * q + 1) * 64]};
        }
    }
    const int t_begin = w.c * kRT;
    const int ntiles = min(kRT, L - t_begin) / kRTE;
    const float* bcp = g.bc + ((int64_t)(w.b * p.n_groups + w.grp) * g.lpad + t_begin) * 32;
    f16v Bq0, Cq0, Bq1, Cq1;
    VMS_SLOAD16(Bq0, bcp, 0);
    VMS_SLOAD16(Cq0, bcp, 0x40);
    auto pt = [&](int tile) { const int t0 = t_begin + tile * kRTE; return REV ? L - t0 - kRTE : t0; };
    Tile<T>::dma_in(u0, p.u_d_stride, pt(0), img_u, lane);
    Tile<T>::dma_in(d0, p.delta_d_stride, pt(0), img_d, lane);
    if (HZ) Tile<T>::dma_in(z0, p.z_d_stride, pt(0), img_z, lane);
    VMS_WAIT_VM();
    for (int tile = 0; tile < ntiles; ++tile) {
        Raw16<T, REV> ru, rd, rz;
        ru.read(img_u, lane);
        rd.read(img_d, lane);
        if (HZ) rz.read(img_z, lane);
        VMS_WAIT_LGKM();
#ifndef VMS_DBG_NOVMEM
        if (tile + 1 < ntiles) {
            Tile<T>::dma_in(u0, p.u_d_stride, pt(tile + 1), img_u, lane);
            Tile<T>::dma_in(d0, p.delta_d_stride, pt(tile + 1), img_d, lane);
            if (HZ) Tile<T>::dma_in(z0, p.z_d_stride, pt(tile + 1), img_z, lane);
        }
#endif
#pragma unroll
        for (int h = 0; h < kRTE / 8; ++h) {
            f2 d2[4], v2[4], y2[kRE];
#pragma unroll
            for (int i = 0; i < kRE; ++i) {
                float t = rd.at(h * 8 + i) + bias;
                if (SP) t = softplusf_(t);
                const float uv = ru.at(h * 8 + i);
                d2[i / 2][i % 2] = t;
                v2[i / 2][i % 2] = t * uv;
                y2[i] = f2{Dd * uv, 0.f};
            }
            VMS_P2_ELEM(0, Bq0, Cq0, Bq1, Cq1, 0x80, 0xc0)
            VMS_P2_ELEM(1, Bq1, Cq1, Bq0, Cq0, 0x100, 0x140)
            VMS_P2_ELEM(2, Bq0, Cq0, Bq1, Cq1, 0x180, 0x1c0)
            VMS_P2_ELEM(3, Bq1, Cq1, Bq0, Cq0, 0x200, 0x240)
            VMS_P2_ELEM(4, Bq0, Cq0, Bq1, Cq1, 0x280, 0x2c0)
            VMS_P2_ELEM(5, Bq1, Cq1, Bq0, Cq0, 0x300, 0x340)
            VMS_P2_ELEM(6, Bq0, Cq0, Bq1, Cq1, 0x380, 0x3c0)
            VMS_P2_ELEM(7, Bq1, Cq1, Bq0, Cq0, 0x400, 0x440)
            bcp += kRE * 32;
            float y[kRE];
#pragma unroll
            for (int i = 0; i < kRE; ++i) y[i] = y2[i].x + y2[i].y;
            stage8<T, REV>(img_o, lane, h, y);
            if (HZ) {
#pragma unroll
                for (int i = 0; i < kRE; ++i) {
                    const float zv = rz.at(h * 8 + i);
                    y[i] *= zv * sigmoidf_(zv);
                }
                stage8<T, REV>(img_oz, lane, h, y);
            }
        }
        VMS_LDS_ORDER();
#ifdef VMS_DBG_NOVMEM
        if (tile == ntiles - 1)
#endif
        {
        Tile<T>::flush_out(o0, p.out_d_stride, pt(tile), img_o, lane);
        if (HZ) Tile<T>::flush_out(oz0, p.out_z_d_stride, pt(tile), img_oz, lane);
        }
        // the next tile's DMA is older than these stores: wait for it, not for them
        if (Tile<T>::SEG * (HZ ? 2 : 1) == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (Tile<T>::SEG * (HZ ? 2 : 1) == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (Tile<T>::SEG * (HZ ? 2 : 1) == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    }
    VMS_SWAIT2(Bq0, Cq0, x2[7]);
}
#undef VMS_P2_ELEM
#undef VMS_P2_Q

template <typename T>
static int launch_rows_t(const vms_scan_fwd_params& p, const RowsGeom& g, hipStream_t stream) {
    {
        dim3 grid((g.lpad + 255) / 256, p.batch * p.n_groups), block(256);
        hipLaunchKernelGGL((scan_rows_prep_kernel<T>), grid, block, 0, stream, p, g);
    }
    const int rbq = (g.rbpb + 3) / 4;
    dim3 grid(p.batch * rbq * g.nch), block(256);
    const bool sp = p.delta_softplus != 0, rev = p.reverse != 0, hz = p.z != nullptr;
#define VMS_P1(SP_, R_) hipLaunchKernelGGL((scan_rows_p1_kernel<T, SP_, R_>), grid, block, 0, stream, p, g)
    if (sp) { if (rev) VMS_P1(true, true); else VMS_P1(true, false); }
    else { if (rev) VMS_P1(false, true); else VMS_P1(false, false); }
#undef VMS_P1
    {
        const int64_t threads = (int64_t)p.batch * g.rbpb * 16 * 64;
        hipLaunchKernelGGL(scan_rows_carry_kernel, dim3((threads + 255) / 256), dim3(256), 0, stream, p, g);
    }
#define VMS_P2(Z_, SP_, R_) hipLaunchKernelGGL((scan_rows_p2_kernel<T, Z_, SP_, R_>), grid, block, 0, stream, p, g)
#define VMS_P2Z(SP_, R_) do { if (hz) VMS_P2(true, SP_, R_); else VMS_P2(false, SP_, R_); } while (0)
    if (sp) { if (rev) VMS_P2Z(true, true); else VMS_P2Z(true, false); }
    else { if (rev) VMS_P2Z(false, true); else VMS_P2Z(false, false); }
#undef VMS_P2Z
#undef VMS_P2
    VMS_LAUNCH_CHECK();
    return VMS_OK;
}

// x must be the dense reference-shaped tensor followed by the checkpoint region (x_has_sub == 2)
int launch_scan_fwd_rows(const vms_scan_fwd_params& p, hipStream_t stream) {
    RowsGeom g = rows_geom(p);
    char* ws = static_cast<char*>(p.workspace);
    g.bc = reinterpret_cast<float*>(ws);
    g.agg = reinterpret_cast<float*>(ws + rows_bc_bytes(p, g));
    g.hck = static_cast<float*>(p.x) + (int64_t)p.batch * p.dim * p.n_chunks * (2 * kRN);
    switch (p.dtype) {
        case VMS_F32: return launch_rows_t<float>(p, g, stream);
        case VMS_F16: return launch_rows_t<f16_t>(p, g, stream);
        default: return launch_rows_t<bf16_t>(p, g, stream);
    }
}

}  // namespace vms
