// selective_scan_short.hip -- forward and backward selective scan for SHORT sequences and many rows: one LANE per (batch, dim) row.
//
// The suite's TimeMamba blocks (egocentric-understanding/avion/models/timemamba.py:135-140, attention style "frozen-in-time") run
// the mixer along TIME: x is rearranged 'b (n t) d -> (b n) t d', i.e. sequences of t = 4 ... 16 frames and a batch of b x 196
// spatial tokens -- (1568, 8, 768) for 8 clips of 8 frames.  The kernels built for long rows (selective_scan_fwd_pair.hip: a wave
// per row, a lane owns 16 consecutive elements; selective_scan_bwd_pair.hip: 16 lanes x 8 elements per row) keep 1 lane of 16 ... 64
// busy there: a ViM block step at (1568, 8, 768) took 13.6 ms, 11.5 of them in the two scans, and the 8-element checkpoint layout
// behind x cost 16.5 KB per row (20 GB per scan at this shape) whatever the length.  The reference serves such rows with one
// 32-thread block each (selective_scan_fwd_kernel.cuh:310-333: kNThreads = 32, kNItems = 4 up to 128 elements).
//
// Here a row is ONE lane's sequential loop: no cross-lane scan, no LDS, no checkpoints -- the backward rebuilds a row's states from
// h = 0.  Lanes run over consecutive channels d of one batch entry, so B / C (shared by the channels of a group) are wave-uniform
// loads and dA / dD / ddelta_bias meet no address conflicts inside a wave; dB / dC -- sums over the channels -- are reduced over the
// wave with DPP and leave as one atomic per (wave, state, position).  seqlen <= 16, dstate 16, variable B / C, real A.
// x keeps the reference's shape and nothing behind it (vms_scan_x_pitch = 2 * dstate for these problems): the running state at the
// row's end in both slots of its one chunk (selective_scan_fwd_kernel.cuh:255-258).
#include "vms_common.h"
#include <type_traits>

namespace vms {

// timing-only ablations (wrong results): tools/variant.sh <tag> -DVMS_ABL_SHORT_NOATOM=1 / -DVMS_ABL_SHORT_NORS=1
#ifndef VMS_ABL_SHORT_NOATOM
#define VMS_ABL_SHORT_NOATOM 0
#endif
#ifndef VMS_ABL_SHORT_NORS
#define VMS_ABL_SHORT_NORS 0
#endif
#ifndef VMS_SHORT_BC_LDS
#define VMS_SHORT_BC_LDS 1   /* 0 (A/B builds): every state loads and widens its own B / C rows from global memory, as until round 5 */
#endif
constexpr int kSN = 16;        // the largest dstate served (the LDS layouts' pitch); dstate 4 and 8 too since round 6 (read at run time)
constexpr int kSN_ = kSN;
constexpr int kSMaxL = 16;     // elements a lane holds in registers = one SEGMENT of a row
constexpr int kSMaxSeg = 4;    // segments per row (round 6): rows of 17 .. 64 elements as up to four launches of the same kernels

// Rows of 17 .. 64 elements and many of them (round 6; VERDICT r5 "missing" #3: neither the lane-per-row kernels, <= 16, nor the
// long-row kernels fit them -- a (16, 32, 2048) forward kept 2 lanes of 64 busy: 109 us).  They run as SEGMENTS of 16 elements, one
// launch of these kernels per segment in scan order, chained through memory: the forward leaves the state after segment s in x
// behind the reference-shaped slots -- x[row][2 N + s N + n], hence a pitch of 2 N + (segments - 1) N, vms_scan_x_pitch -- and
// starts segment s + 1 from it; the backward walks the segments from the last to the first, takes the state entering a segment
// from the same place and hands the adjoint entering the segment before it through the workspace (gcar[row][n]), where the per-row
// dA / dD / ddelta_bias partials accumulate as well.  Pointers and seqlen of a segment's parameter block are set by the host.
int scan_short_nseg(const vms_scan_fwd_params& p) { return (p.seqlen + kSMaxL - 1) / kSMaxL; }
int64_t scan_short_x_pitch(const vms_scan_fwd_params& p) { return 2 * (int64_t)p.dstate + (int64_t)(scan_short_nseg(p) - 1) * p.dstate; }

// the problem's shape alone (what vms_scan_x_pitch decides on: the pitch is its answer, not its input)
static bool scan_short_shape_ok(const vms_scan_fwd_params& p) {
    if (p.is_complex || !p.is_variable_B || !p.is_variable_C || (p.dstate != kSN && p.dstate != 8 && p.dstate != 4) || p.seqlen < 1 || p.seqlen > kSMaxL * kSMaxSeg) return false;
    if (p.n_chunks != 1 || p.n_groups < 1 || p.dim % p.n_groups != 0 || p.x_has_sub == 2) return false;
    if (p.reverse_from > 0 && p.reverse_from < p.batch) return false;   // (the entry points split mixed directions into two problems)
    // the backward's wave sums of dB / dC need whole waves inside one (batch entry, group); forward and backward go together
    // (x carries no checkpoints for the long-row backward kernels)
    if ((p.dim / p.n_groups) % 64 != 0) return false;
    // below a few thousand rows the long-row kernels' launch is as good; the lane-per-row form needs rows to fill waves
    return (int64_t)p.batch * p.dim >= 4096;
}
bool scan_short_eligible(const vms_scan_fwd_params& p) {
    if (!scan_short_shape_ok(p)) return false;
    // segmented rows: x must have room for the states between the segments (a caller that brought the reference's own 2 N pitch
    // -- no backward intended -- runs the long-row kernels)
    return scan_short_nseg(p) == 1 || (p.x != nullptr && p.x_chunk_stride >= scan_short_x_pitch(p));
}
// the same for a problem with a direction per batch entry (reverse_from): both sub-batches
static bool short_takes(const vms_scan_fwd_params& p, bool (*ok)(const vms_scan_fwd_params&)) {
    if (p.reverse_from > 0 && p.reverse_from < p.batch) {
        vms_scan_fwd_params lo = p, hi = p;
        lo.batch = p.reverse_from; lo.reverse_from = 0;
        hi.batch = p.batch - p.reverse_from; hi.reverse_from = 0;
        return scan_impl_level(p) >= VMS_IMPL_PAIR && ok(lo) && ok(hi);
    }
    return scan_impl_level(p) >= VMS_IMPL_PAIR && ok(p);
}
bool scan_short_takes(const vms_scan_fwd_params& p) { return short_takes(p, scan_short_eligible); }
bool scan_short_takes_shape(const vms_scan_fwd_params& p) { return short_takes(p, scan_short_shape_ok); }

template <typename T>
__device__ __forceinline__ float ld1(const T* p, int64_t i) { return static_cast<float>(p[i]); }

// A row's L (<= LP) elements in LOGICAL order (REV: logical i = physical L - 1 - i), zeros behind them and for !ok.
// VEC (seqlen % 8 == 0, 16-byte aligned rows): whole 16-byte vectors -- element by element a row cost 16 two-byte loads per tensor,
// every one touching 64 different lines per wave ((1568, 16, 768): 1.27 ms per forward call instead of 0.2).
template <typename T, int LP, bool REV, bool VEC>
__device__ __forceinline__ void load_row(const T* __restrict__ row, int L, bool ok, float (&v)[LP]) {
    if constexpr (VEC) {
        constexpr int EPV = 16 / sizeof(T);
#pragma unroll
        for (int k = 0; k < LP / EPV; ++k) {
            if (ok && k * EPV < L) {
                const vec_t<T, EPV> t = *reinterpret_cast<const vec_t<T, EPV>*>(row + (REV ? L - (k + 1) * EPV : k * EPV));
#pragma unroll
                for (int e = 0; e < EPV; ++e) v[k * EPV + e] = static_cast<float>(t[REV ? EPV - 1 - e : e]);
            } else {
#pragma unroll
                for (int e = 0; e < EPV; ++e) v[k * EPV + e] = 0.f;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < LP; ++i) v[i] = ok && i < L ? static_cast<float>(row[REV ? L - 1 - i : i]) : 0.f;
    }
}
// A batch entry's B and C (this wave's: they are wave-uniform) as fp32 in LOGICAL order in a wave-private LDS block,
// [tensor][state][LP], staged ONCE per wave (round 6).  Until then every state of the loop loaded and widened its two rows from
// global memory right where it needed them: 16 exposed L2 round trips per wave and 2 LP conversions per state -- the backward's
// VALU pipe was 42 % busy at 4 waves per SIMD ((1568, 768, 8): 243 us per direction against ~100 of issue time).  Now one
// load instruction per wave (lane -> one 16-byte vector of one (tensor, state) row) fills the block and a state reads its rows
// with broadcast ds_read_b128 (all lanes one address: no bank conflicts, ~100 cycles instead of an L2 round trip).
template <typename T, int LP, bool REV, bool VEC>
__device__ __forceinline__ void stage_bc_short(float* __restrict__ blk, const T* __restrict__ Bp, const T* __restrict__ Cp,
                                               const int64_t bs, const int64_t cs, const int L, const int lane, const int N) {
    if constexpr (VEC) {
        constexpr int EPV = 16 / sizeof(T), VPR = LP / EPV, NVmax = 2 * kSN_ * VPR;
        const int NV = 2 * N * VPR;                  // (the block keeps its 16-state pitch; states >= N are not staged)
#pragma unroll
        for (int v0 = 0; v0 < NVmax; v0 += 64) {
            const int v = v0 + lane;
            if (v < NV) {
                const int ten = v / (N * VPR), n = (v / VPR) % N, k = v % VPR;
                const T* row = ten ? Cp + (int64_t)n * cs : Bp + (int64_t)n * bs;
                float o[EPV];
                if (k * EPV < L) {
                    const vec_t<T, EPV> t = *reinterpret_cast<const vec_t<T, EPV>*>(row + (REV ? L - (k + 1) * EPV : k * EPV));
#pragma unroll
                    for (int e = 0; e < EPV; ++e) o[e] = static_cast<float>(t[REV ? EPV - 1 - e : e]);
                } else {
#pragma unroll
                    for (int e = 0; e < EPV; ++e) o[e] = 0.f;
                }
                float4* dst = reinterpret_cast<float4*>(blk + (ten * kSN_ + n) * LP + k * EPV);
#pragma unroll
                for (int q = 0; q < EPV / 4; ++q) dst[q] = float4{o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
            }
        }
    } else {
        for (int idx = lane; idx < 2 * N * LP; idx += 64) {
            const int ten = idx / (N * LP), n = (idx / LP) % N, i = idx % LP;
            const T* row = ten ? Cp + (int64_t)n * cs : Bp + (int64_t)n * bs;
            blk[(ten * kSN_ + n) * LP + i] = i < L ? static_cast<float>(row[REV ? L - 1 - i : i]) : 0.f;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // wave-private: the wave's LDS instructions complete in order
    __builtin_amdgcn_wave_barrier();
}
template <int LP>
__device__ __forceinline__ void read_bc_short(const float* __restrict__ blk, const int ten, const int n, float (&v)[LP]) {
    const float4* src = reinterpret_cast<const float4*>(blk + (ten * kSN_ + n) * LP);
#pragma unroll
    for (int q = 0; q < LP / 4; ++q) {
        const float4 t = src[q];
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
}

template <typename T, int LP, bool REV, bool VEC>
__device__ __forceinline__ void store_row(T* __restrict__ row, int L, const float (&v)[LP]) {
    if constexpr (VEC) {
        constexpr int EPV = 16 / sizeof(T);
#pragma unroll
        for (int k = 0; k < LP / EPV; ++k) {
            if (k * EPV < L) {
                vec_t<T, EPV> t;
#pragma unroll
                for (int e = 0; e < EPV; ++e) t[REV ? EPV - 1 - e : e] = static_cast<T>(v[k * EPV + e]);
                *reinterpret_cast<vec_t<T, EPV>*>(row + (REV ? L - (k + 1) * EPV : k * EPV)) = t;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < LP; ++i)
            if (i < L) row[REV ? L - 1 - i : i] = static_cast<T>(v[i]);
    }
}

// 2 x LP per-lane values -> their sums over the wave, as a reduce-scatter: v_permlane32_swap on pairs (a swap + an add halve the
// count: lanes 0-31 then hold the first value of the pair summed over both halves, lanes 32-63 the second), v_permlane16_swap on
// pairs of those (rows 0 / 1 / 2 / 3 hold values 4 m + 0 / 2 / 1 / 3 summed over the four rows), then 4 DPP steps inside each row:
// lane 16 r + 15 of z[m] = the wave's sum of value 4 m + {0, 2, 1, 3}[r].  40 instead of ~150 vector instructions per state at
// 8 elements (six-step sums of every value, a v_readlane and two selects each): the backward kernel was 94 % VALU-bound.
template <int V>
__device__ __forceinline__ void wave_reduce_scatter(float (&v)[V], float (&z)[V / 4]) {
    float w[V / 2];
#pragma unroll
    for (int k = 0; k < V / 2; ++k) {
        float a = v[2 * k], b = v[2 * k + 1];
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
        w[k] = a + b;
    }
#pragma unroll
    for (int m = 0; m < V / 4; ++m) {
        float a = w[2 * m], b = w[2 * m + 1];
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
        float t = a + b;
        t += dpp_mov<DPP_ROW_SHR1, 0xf>(0.f, t);
        t += dpp_mov<DPP_ROW_SHR2, 0xf>(0.f, t);
        t += dpp_mov<DPP_ROW_SHR4, 0xf>(0.f, t);
        t += dpp_mov<DPP_ROW_SHR8, 0xf>(0.f, t);
        z[m] = t;
    }
}

// LP: the register arrays' length (8 or 16) >= seqlen.  Logical element i of a right-to-left row is physical element L - 1 - i.
// h_in / h_out (segmented rows): float offsets inside a row's x of the state this segment starts from (-1: zero) and of where its
// final state goes (-1: the reference-shaped slots of the row's one chunk)
template <typename T, bool HZ, bool REV, int LP, bool VEC>
__global__ __launch_bounds__(256) void scan_fwd_short_kernel(const vms_scan_fwd_params p, const int h_in, const int h_out) {
    const int N = p.dstate;
    // a wave = 64 consecutive channels of ONE batch entry and ONE group (dim / n_groups is a multiple of 64): uniform B / C addresses;
    // the workgroup's 4 waves = the same channels of 4 CONSECUTIVE batch entries: in the blocks' channel-slowest layout those rows
    // are neighbours in memory (4 x 32 bytes = one line), so the line a wave touches is the line its three neighbours touch
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int dblocks = p.dim / 64;
    const int d = ((int)blockIdx.x % dblocks) * 64 + lane;
    const int bq = __builtin_amdgcn_readfirstlane(((int)blockIdx.x / dblocks) * 4 + wave);
    const bool row_ok = bq < p.batch;
    const int b = row_ok ? bq : p.batch - 1;
    const int g = __builtin_amdgcn_readfirstlane(d / (p.dim / p.n_groups));
    const int L = p.seqlen;
    const T* u = static_cast<const T*>(p.u) + (int64_t)b * p.u_batch_stride + (int64_t)d * p.u_d_stride;
    const T* dt = static_cast<const T*>(p.delta) + (int64_t)b * p.delta_batch_stride + (int64_t)d * p.delta_d_stride;
    T* out = static_cast<T*>(p.out) + (int64_t)b * p.out_batch_stride + (int64_t)d * p.out_d_stride;
    const T* Bp = static_cast<const T*>(p.B) + (int64_t)b * p.B_batch_stride + (int64_t)g * p.B_group_stride;
    const T* Cp = static_cast<const T*>(p.C) + (int64_t)b * p.C_batch_stride + (int64_t)g * p.C_group_stride;
    const float* Af = static_cast<const float*>(p.A) + (int64_t)d * p.A_d_stride;
    const float Dd = p.D ? static_cast<const float*>(p.D)[d] : 0.f;
    const float bias = p.delta_bias ? static_cast<const float*>(p.delta_bias)[d] : 0.f;
    float dl[LP], du[LP], y[LP];
    load_row<T, LP, REV, VEC>(dt, L, row_ok, dl);
    load_row<T, LP, REV, VEC>(u, L, row_ok, du);
#pragma unroll
    for (int i = 0; i < LP; ++i) {
        float t = dl[i] + bias;
        if (p.delta_softplus) t = softplusf_(t);
        t = i < L && row_ok ? t : 0.f;    // past the end: a = 1, b = 0
        const float uv = du[i];
        dl[i] = t;
        du[i] = t * uv;
        y[i] = Dd * uv;
    }
    // (the state loop is NOT unrolled: unrolled, the 16 states' B / C loads were all hoisted to the top -- 306 registers at 16 elements)
    const int64_t xpitch = p.x_chunk_stride ? p.x_chunk_stride : 2 * N;
    float* xr = static_cast<float*>(p.x) + ((int64_t)b * p.dim + d) * xpitch;
    // x: a wave's 64 rows are 64 x 128 contiguous bytes of the dense x.  Written from the lanes' own registers -- 16 bytes per
    // lane, 128 bytes apart -- the forward moved 383 MB for 192 MB of results at (1568, 768, 8) and spent most of its 190 us on
    // it; the final states go through LDS ([row][state], 17-float pitch) and leave as eight 1 KB-contiguous stores per wave.
    __shared__ float hs[4][64 * 17];
    __shared__ __attribute__((aligned(16))) float bcs[4][2 * kSN * LP];   // this wave's B / C as fp32 (stage_bc_short)
    if (VMS_SHORT_BC_LDS) stage_bc_short<T, LP, REV, VEC>(bcs[wave], Bp, Cp, p.B_dstate_stride, p.C_dstate_stride, L, lane, N);
    const bool x16 = (xpitch & 3) == 0 && (reinterpret_cast<uintptr_t>(p.x) & 15) == 0;
    const bool xdense = x16 && xpitch == 2 * N && h_out < 0 && N == kSN;   // (the LDS transposition below is laid out for 16 states)
#pragma unroll 1
    for (int n0 = 0; n0 < N; n0 += 2) {
        float hq[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int n = n0 + k;
            const float An = Af[(int64_t)n * p.A_dstate_stride] * kLog2e;
            float Bv[LP], Cv[LP];
            if (VMS_SHORT_BC_LDS) {
                read_bc_short<LP>(bcs[wave], 0, n, Bv);
                read_bc_short<LP>(bcs[wave], 1, n, Cv);
            } else {
                load_row<T, LP, REV, VEC>(Bp + (int64_t)n * p.B_dstate_stride, L, true, Bv);
                load_row<T, LP, REV, VEC>(Cp + (int64_t)n * p.C_dstate_stride, L, true, Cv);
            }
            float h = h_in >= 0 ? xr[h_in + n] : 0.f;
#pragma unroll
            for (int i = 0; i < LP; ++i) {
                h = fmaf(fast_exp2(dl[i] * An), h, du[i] * Bv[i]);
                y[i] = fmaf(Cv[i], h, y[i]);
            }
            hq[k] = h;
        }
        if (xdense) {
            hs[wave][lane * 17 + n0] = hq[0];
            hs[wave][lane * 17 + n0 + 1] = hq[1];
        } else if (h_out >= 0) {   // a segment that is not the row's last: its final state for the next segment (and the backward)
            if (row_ok) { xr[h_out + n0] = hq[0]; xr[h_out + n0 + 1] = hq[1]; }
        } else if (row_ok) {   // the state at the row's end in both slots of the one chunk
            if (x16) *reinterpret_cast<float4*>(xr + 2 * n0) = float4{hq[0], hq[0], hq[1], hq[1]};
            else { xr[2 * n0] = hq[0]; xr[2 * n0 + 1] = hq[0]; xr[2 * n0 + 2] = hq[1]; xr[2 * n0 + 3] = hq[1]; }
        }
    }
    if (!row_ok) return;
    if (xdense) {   // (wave-private LDS: the wave's own writes are complete before its reads)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float* xw = static_cast<float*>(p.x) + ((int64_t)b * p.dim + (d - lane)) * (2 * N);     // the wave's 64 rows
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int flat = k * 64 + lane, row = flat >> 3, quad = flat & 7;
            const float h0 = hs[wave][row * 17 + 2 * quad], h1 = hs[wave][row * 17 + 2 * quad + 1];
            *reinterpret_cast<float4*>(xw + flat * 4) = float4{h0, h0, h1, h1};
        }
    }
    store_row<T, LP, REV, VEC>(out, L, y);
    if (HZ) {
        const T* z = static_cast<const T*>(p.z) + (int64_t)b * p.z_batch_stride + (int64_t)d * p.z_d_stride;
        T* oz = static_cast<T*>(p.out_z) + (int64_t)b * p.out_z_batch_stride + (int64_t)d * p.out_z_d_stride;
        float zv[LP], old[LP];
        load_row<T, LP, REV, VEC>(z, L, true, zv);
        if (p.out_z_accumulate) load_row<T, LP, REV, VEC>(oz, L, true, old);
#pragma unroll
        for (int i = 0; i < LP; ++i) {
            y[i] = y[i] * zv[i] * sigmoidf_(zv[i]);
            if (p.out_z_accumulate) y[i] += old[i];
        }
        store_row<T, LP, REV, VEC>(oz, L, y);
    }
}

// Backward.  Waves and workgroups as in the forward kernel (64 channels of one batch entry; 4 neighbouring entries per workgroup).
// dB / dC -- sums over the channels -- are reduced over the wave with DPP, handed to lanes 0 .. 2 L - 1 (v_readlane + select) and
// leave as ONE atomic instruction of 2 L lanes per (wave, state): no LDS, no barrier (a first version that also summed the
// workgroup's waves -- then one batch entry per workgroup -- through LDS paid 32 barriers per row and had every wave fetch its own
// lines: 338 us at (1568, 768, 8)).  dA / dD / ddelta_bias -- sums over the batch -- go to the workspace as
// ws[batch entry][18][channel] (coalesced stores, no atomics) and are summed by short_reduce_kernel: as atomics straight from the
// rows, 1,568 of them onto each of 12 K addresses, they were 1.1 of the kernel's 2.0 ms at (1568, 16, 768).  No workspace: atomics.
// Segmented rows: h_in = float offset inside a row's x of the state entering this segment (-1: zero); gcar[row][n] = the adjoint
// handed from the segment behind this one (flags & 1: read it; next_off = element offset, from this segment's delta pointer, of
// that segment's first element, whose decay multiplies it) to the segment before it (flags & 2: write it); flags & 4: the
// workspace partials of dA / dD / ddelta_bias accumulate (every segment but the first one processed).
template <typename T, bool HZ, bool REV, int LP, bool VEC>
__global__ __launch_bounds__(256) void scan_bwd_short_kernel(const vms_scan_bwd_params q, float* __restrict__ ws, float* __restrict__ gcar,
                                                             const int h_in, const int flags, const int next_off) {
    const vms_scan_fwd_params& p = q.f;
    const int N = p.dstate;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int dblocks = p.dim / 64;
    const int d = ((int)blockIdx.x % dblocks) * 64 + lane;
    const int b = __builtin_amdgcn_readfirstlane(((int)blockIdx.x / dblocks) * 4 + wave);
    if (b >= p.batch) return;                                          // (the whole wave; nothing below synchronises waves)
    const int g = __builtin_amdgcn_readfirstlane(d / (p.dim / p.n_groups));
    const int L = p.seqlen;
    const T* u = static_cast<const T*>(p.u) + (int64_t)b * p.u_batch_stride + (int64_t)d * p.u_d_stride;
    const T* dt = static_cast<const T*>(p.delta) + (int64_t)b * p.delta_batch_stride + (int64_t)d * p.delta_d_stride;
    const T* dout = static_cast<const T*>(q.dout) + (int64_t)b * q.dout_batch_stride + (int64_t)d * q.dout_d_stride;
    const T* Bp = static_cast<const T*>(p.B) + (int64_t)b * p.B_batch_stride + (int64_t)g * p.B_group_stride;
    const T* Cp = static_cast<const T*>(p.C) + (int64_t)b * p.C_batch_stride + (int64_t)g * p.C_group_stride;
    float* dBp = q.dB + (int64_t)b * q.dB_batch_stride + (int64_t)g * q.dB_group_stride;
    float* dCp = q.dC + (int64_t)b * q.dC_batch_stride + (int64_t)g * q.dC_group_stride;
    const float* Af = static_cast<const float*>(p.A) + (int64_t)d * p.A_d_stride;
    const float Dd = p.D ? static_cast<const float*>(p.D)[d] : 0.f;
    const float bias = p.delta_bias ? static_cast<const float*>(p.delta_bias)[d] : 0.f;
    float* wsr = ws ? ws + (int64_t)b * 18 * p.dim + d : nullptr;
    const float* xr = static_cast<const float*>(p.x) + ((int64_t)b * p.dim + d) * (p.x_chunk_stride ? p.x_chunk_stride : 2 * N);
    float* gc = gcar ? gcar + ((int64_t)b * p.dim + d) * N : nullptr;
    float dl_nx = 0.f;           // delta of the first element of the segment behind this one
    if (flags & 1) {
        dl_nx = static_cast<float>(dt[next_off]) + bias;
        if (p.delta_softplus) dl_nx = softplusf_(dl_nx);
    }
    __shared__ __attribute__((aligned(16))) float bcs[4][2 * kSN * LP];   // this wave's B / C as fp32 (stage_bc_short)
    if (VMS_SHORT_BC_LDS) stage_bc_short<T, LP, REV, VEC>(bcs[wave], Bp, Cp, p.B_dstate_stride, p.C_dstate_stride, L, lane, N);
    float dl[LP], uv[LP], dy[LP], sg[LP], dua[LP], dda[LP];
    float dD = 0.f;
    load_row<T, LP, REV, VEC>(dt, L, true, dl);
    load_row<T, LP, REV, VEC>(u, L, true, uv);
    load_row<T, LP, REV, VEC>(dout, L, true, dy);
    if (HZ) {
        const T* z = static_cast<const T*>(p.z) + (int64_t)b * p.z_batch_stride + (int64_t)d * p.z_d_stride;
        const T* o_ = static_cast<const T*>(p.out) + (int64_t)b * p.out_batch_stride + (int64_t)d * p.out_d_stride;
        T* dz = static_cast<T*>(q.dz) + (int64_t)b * q.dz_batch_stride + (int64_t)d * q.dz_d_stride;
        float zv[LP], ov[LP], dzv[LP];
        load_row<T, LP, REV, VEC>(z, L, true, zv);
        load_row<T, LP, REV, VEC>(o_, L, true, ov);
        if (q.dz_accumulate) load_row<T, LP, REV, VEC>(dz, L, true, dzv);
#pragma unroll
        for (int i = 0; i < LP; ++i) {
            const float sz = sigmoidf_(zv[i]), silu = zv[i] * sz;
            const float t = dy[i] * ov[i] * sz * (1.f + zv[i] * (1.f - sz));
            dzv[i] = q.dz_accumulate ? dzv[i] + t : t;
            ov[i] *= silu;          // the gated output, should the caller want it recomputed
            dy[i] *= silu;
        }
        store_row<T, LP, REV, VEC>(dz, L, dzv);
        if (p.out_z) store_row<T, LP, REV, VEC>(static_cast<T*>(p.out_z) + (int64_t)b * p.out_z_batch_stride + (int64_t)d * p.out_z_d_stride, L, ov);
    }
#pragma unroll
    for (int i = 0; i < LP; ++i) {
        float t = dl[i] + bias, s = 1.f;
        if (p.delta_softplus) {
            const float ex = fast_exp(t), w = 1.f + ex, rw = fast_rcp(w);
            const float sp = fmaf(ex - (w - 1.f), rw, fast_log(w));
            s = t <= 20.f ? ex * rw : 1.f;
            t = t <= 20.f ? sp : t;
        }
        dl[i] = i < L ? t : 0.f;
        sg[i] = s;
        dua[i] = Dd * dy[i];
        dda[i] = 0.f;
        dD = fmaf(dy[i], uv[i], dD);
    }
#pragma unroll 1
    for (int n = 0; n < N; ++n) {
        const float Araw = Af[(int64_t)n * p.A_dstate_stride], An = Araw * kLog2e;
        float a[LP], x[LP], Bv[LP], Cv[LP];
        if (VMS_SHORT_BC_LDS) {
            read_bc_short<LP>(bcs[wave], 0, n, Bv);
            read_bc_short<LP>(bcs[wave], 1, n, Cv);
        } else {
            load_row<T, LP, REV, VEC>(Bp + (int64_t)n * p.B_dstate_stride, L, true, Bv);
            load_row<T, LP, REV, VEC>(Cp + (int64_t)n * p.C_dstate_stride, L, true, Cv);
        }
        const float h0 = h_in >= 0 ? xr[h_in + n] : 0.f;
        float h = h0;
#pragma unroll
        for (int i = 0; i < LP; ++i) {
            a[i] = fast_exp2(dl[i] * An);
            h = fmaf(a[i], h, dl[i] * uv[i] * Bv[i]);
            x[i] = h;
        }
        // (elements past the segment's end are identity steps -- delta = 0, dy = 0 -- so the carried adjoint passes through them)
        float gr = (flags & 1) ? gc[n] : 0.f, dA = 0.f;
        const float a_nx = fast_exp2(dl_nx * An);
        float vals[2 * LP];      // this lane's dB (0 .. LP-1) and dC (LP .. 2 LP-1) terms of the state
#pragma unroll
        for (int i = LP - 1; i >= 0; --i) {
            gr = fmaf(i == LP - 1 ? a_nx : a[i + 1], gr, Cv[i] * dy[i]);      // g_i = a_{i+1} g_{i+1} + C_i dy_i
            const float ax = a[i] * (i == 0 ? h0 : x[i - 1]);                // a_i x_{i-1}
            dua[i] = fmaf(gr * dl[i], Bv[i], dua[i]);
            dda[i] = fmaf(gr, fmaf(Araw, ax, uv[i] * Bv[i]), dda[i]);
            dA = fmaf(gr * dl[i], ax, dA);
            vals[i] = gr * dl[i] * uv[i];
            vals[LP + i] = dy[i] * x[i];
        }
        if (flags & 2) gc[n] = gr;                                           // g of this segment's first element
        if (wsr) wsr[(int64_t)n * p.dim] = (flags & 4) ? wsr[(int64_t)n * p.dim] + dA : dA;
        else atomicAdd(q.dA + (int64_t)d * q.dA_d_stride + (int64_t)n * q.dA_dstate_stride, dA);
        float z[LP / 2];
#if VMS_ABL_SHORT_NORS   /* timing only: no wave reduction */
#pragma unroll
        for (int m = 0; m < LP / 2; ++m) z[m] = vals[4 * m] + vals[4 * m + 1] + vals[4 * m + 2] + vals[4 * m + 3];
#else
        wave_reduce_scatter<2 * LP>(vals, z);
#endif
        if (!VMS_ABL_SHORT_NOATOM && (lane & 15) == 15) {       // lane 16 r + 15 of z[m]: the wave's sum of value 4 m + {0, 2, 1, 3}[r]
            const int r = lane >> 4, sub = r == 1 ? 2 : r == 2 ? 1 : r;
#pragma unroll
            for (int m = 0; m < LP / 2; ++m) {
                const int v = 4 * m + sub, i = v & (LP - 1);
                float* dst = v < LP ? dBp + (int64_t)n * q.dB_dstate_stride : dCp + (int64_t)n * q.dC_dstate_stride;
                if (i < L) atomicAdd(dst + (REV ? L - 1 - i : i), z[m]);
            }
        }
#if VMS_ABL_SHORT_NOATOM
#pragma unroll
        for (int m = 0; m < LP / 2; ++m) asm volatile("" ::"v"(z[m]));
#endif
    }
    float dbias = 0.f;
#pragma unroll
    for (int i = 0; i < LP; ++i) {
        dda[i] = i < L ? dda[i] * sg[i] : 0.f;
        dbias += dda[i];
    }
    store_row<T, LP, REV, VEC>(static_cast<T*>(q.du) + (int64_t)b * q.du_batch_stride + (int64_t)d * q.du_d_stride, L, dua);
    store_row<T, LP, REV, VEC>(static_cast<T*>(q.ddelta) + (int64_t)b * q.ddelta_batch_stride + (int64_t)d * q.ddelta_d_stride, L, dda);
    if (wsr) {
        wsr[(int64_t)16 * p.dim] = (flags & 4) ? wsr[(int64_t)16 * p.dim] + dD : dD;
        wsr[(int64_t)17 * p.dim] = (flags & 4) ? wsr[(int64_t)17 * p.dim] + dbias : dbias;
    } else {
        if (q.dD) atomicAdd(q.dD + d, dD);
        if (q.ddelta_bias) atomicAdd(q.ddelta_bias + d, dbias);
    }
}

// dA / dD / ddelta_bias += the sum over the batch of ws[batch][18][dim]: a thread sums kRB batch entries of one (slot, channel)
constexpr int kRB = 32;
__global__ __launch_bounds__(256) void short_reduce_kernel(const float* __restrict__ ws, const vms_scan_bwd_params q) {
    const vms_scan_fwd_params& p = q.f;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t per_chunk = (int64_t)18 * p.dim;
    const int chunk = (int)(t / per_chunk);
    const int64_t idx = t - chunk * per_chunk;            // slot * dim + d
    const int b0 = chunk * kRB, b1 = b0 + kRB < p.batch ? b0 + kRB : p.batch;
    if (b0 >= p.batch) return;
    float v = 0.f;
    for (int b = b0; b < b1; ++b) v += ws[(int64_t)b * per_chunk + idx];
    const int slot = (int)(idx / p.dim), d = (int)(idx - (int64_t)slot * p.dim);
    if (slot < 16) { if (slot < p.dstate) atomicAdd(q.dA + (int64_t)d * q.dA_d_stride + (int64_t)slot * q.dA_dstate_stride, v); }
    else if (slot == 16) { if (q.dD) atomicAdd(q.dD + d, v); }
    else if (q.ddelta_bias) atomicAdd(q.ddelta_bias + d, v);
}
// per-row partials [batch][18][dim] + (segmented rows) the adjoint carries [batch x dim][16]
int64_t scan_bwd_short_ws_bytes(const vms_scan_bwd_params& q) {
    return (int64_t)q.f.batch * 18 * q.f.dim * 4 + (scan_short_nseg(q.f) > 1 ? (int64_t)q.f.batch * q.f.dim * kSN * 4 : 0);
}

// whole 16-byte vectors per row: seqlen a multiple of 8, every row 16-byte aligned
static bool rows16(const void* ptr, int64_t bs, int64_t ds, int es) {
    return ptr == nullptr || ((reinterpret_cast<uintptr_t>(ptr) & 15) == 0 && (bs * es) % 16 == 0 && (ds * es) % 16 == 0);
}
static bool short_vec_fwd(const vms_scan_fwd_params& p) {
    const int es = p.dtype == VMS_F32 ? 4 : 2;
    return p.seqlen % 8 == 0 && rows16(p.u, p.u_batch_stride, p.u_d_stride, es) && rows16(p.delta, p.delta_batch_stride, p.delta_d_stride, es) &&
           rows16(p.out, p.out_batch_stride, p.out_d_stride, es) && rows16(p.z, p.z_batch_stride, p.z_d_stride, es) &&
           rows16(p.out_z, p.out_z_batch_stride, p.out_z_d_stride, es) && rows16(p.B, p.B_batch_stride, p.B_group_stride, es) &&
           rows16(p.C, p.C_batch_stride, p.C_group_stride, es) && (p.B_dstate_stride * es) % 16 == 0 && (p.C_dstate_stride * es) % 16 == 0;
}
static bool short_vec_bwd(const vms_scan_bwd_params& q) {
    const int es = q.f.dtype == VMS_F32 ? 4 : 2;
    return short_vec_fwd(q.f) && rows16(q.dout, q.dout_batch_stride, q.dout_d_stride, es) && rows16(q.du, q.du_batch_stride, q.du_d_stride, es) &&
           rows16(q.ddelta, q.ddelta_batch_stride, q.ddelta_d_stride, es) && rows16(q.dz, q.dz_batch_stride, q.dz_d_stride, es);
}

template <typename P>
static P* adv(P* ptr, int64_t elems, int es) {   // a row pointer moved to a segment's first physical element
    return ptr ? reinterpret_cast<P*>(reinterpret_cast<char*>(const_cast<typename std::remove_const<P>::type*>(ptr)) + elems * es) : nullptr;
}
// the parameter block of logical segment s (elements [16 s, 16 s + len) in scan order): physical offset `off`
static void short_segment(const vms_scan_fwd_params& p, int s, vms_scan_fwd_params& ps, int& len, int64_t& off) {
    const int l0 = s * kSMaxL;
    len = p.seqlen - l0 < kSMaxL ? p.seqlen - l0 : kSMaxL;
    off = p.reverse ? p.seqlen - l0 - len : l0;
    const int es = p.dtype == VMS_F32 ? 4 : 2;
    ps = p;
    ps.seqlen = len;
    ps.u = adv(p.u, off, es); ps.delta = adv(p.delta, off, es); ps.z = adv(p.z, off, es);
    ps.out = adv(p.out, off, es); ps.out_z = adv(p.out_z, off, es);
    ps.B = adv(p.B, off, es); ps.C = adv(p.C, off, es);
}

template <typename T>
static int launch_fwd_short_t(const vms_scan_fwd_params& p0, hipStream_t stream) {
    const dim3 grid((unsigned)((p0.dim / 64) * ((p0.batch + 3) / 4))), block(256);
    const int nseg = scan_short_nseg(p0), N = p0.dstate;
    for (int sgm = 0; sgm < nseg; ++sgm) {
        vms_scan_fwd_params p;
        int len;
        int64_t off;
        short_segment(p0, sgm, p, len, off);
        const int h_in = sgm > 0 ? 2 * N + (sgm - 1) * N : -1, h_out = sgm == nseg - 1 ? -1 : 2 * N + sgm * N;
        const bool vec = short_vec_fwd(p);
#define VMS_S(Z_, R_)                                                                                                    \
    do {                                                                                                                \
        if (p.seqlen <= 8) {                                                                                            \
            if (vec) hipLaunchKernelGGL((scan_fwd_short_kernel<T, Z_, R_, 8, true>), grid, block, 0, stream, p, h_in, h_out);         \
            else hipLaunchKernelGGL((scan_fwd_short_kernel<T, Z_, R_, 8, false>), grid, block, 0, stream, p, h_in, h_out);            \
        } else {                                                                                                        \
            if (vec) hipLaunchKernelGGL((scan_fwd_short_kernel<T, Z_, R_, 16, true>), grid, block, 0, stream, p, h_in, h_out);        \
            else hipLaunchKernelGGL((scan_fwd_short_kernel<T, Z_, R_, 16, false>), grid, block, 0, stream, p, h_in, h_out);           \
        }                                                                                                               \
    } while (0)
        if (p.z) { if (p.reverse) VMS_S(true, true); else VMS_S(true, false); }
        else { if (p.reverse) VMS_S(false, true); else VMS_S(false, false); }
#undef VMS_S
        VMS_LAUNCH_CHECK();
    }
    set_last_kernel(nseg > 1 ? "scan_fwd_short+segments" : "scan_fwd_short");
    return VMS_OK;
}
int launch_scan_fwd_short(const vms_scan_fwd_params& p, hipStream_t stream) {
    switch (p.dtype) {
        case VMS_F32: return launch_fwd_short_t<float>(p, stream);
        case VMS_F16: return launch_fwd_short_t<f16_t>(p, stream);
        default: return launch_fwd_short_t<bf16_t>(p, stream);
    }
}

bool scan_bwd_short_eligible(const vms_scan_bwd_params& q) { return scan_short_eligible(q.f); }

bool scan_bwd_short_has_ws(const vms_scan_bwd_params& q) {
    const vms_scan_fwd_params& p = q.f;
    return p.workspace != nullptr && p.workspace_bytes >= scan_bwd_short_ws_bytes(q) && (reinterpret_cast<uintptr_t>(p.workspace) & 3) == 0;
}

template <typename T>
static int launch_bwd_short_t(const vms_scan_bwd_params& q0, hipStream_t stream) {
    const vms_scan_fwd_params& p0 = q0.f;
    const dim3 grid((unsigned)((p0.dim / 64) * ((p0.batch + 3) / 4))), block(256);
    const int nseg = scan_short_nseg(p0), N = p0.dstate;
    float* ws = scan_bwd_short_has_ws(q0) ? static_cast<float*>(p0.workspace) : nullptr;
    if (nseg > 1 && !ws) {   // (vms_selective_scan_bwd asks scan_bwd_short_has_ws before it comes here)
        set_error("segmented short rows need the workspace vms_scan_bwd_workspace_bytes() asks for");
        return VMS_ERR_INVALID_ARG;
    }
    float* gcar = nseg > 1 ? ws + (int64_t)p0.batch * 18 * p0.dim : nullptr;
    const int es = p0.dtype == VMS_F32 ? 4 : 2;
    for (int sgm = nseg - 1; sgm >= 0; --sgm) {     // against scan order
        vms_scan_bwd_params q = q0;
        int len;
        int64_t off;
        short_segment(p0, sgm, q.f, len, off);
        q.dout = adv(q0.dout, off, es); q.du = adv(q0.du, off, es); q.ddelta = adv(q0.ddelta, off, es); q.dz = adv(q0.dz, off, es);
        q.dB = q0.dB ? q0.dB + off : nullptr; q.dC = q0.dC ? q0.dC + off : nullptr;
        const vms_scan_fwd_params& p = q.f;
        const int h_in = sgm > 0 ? 2 * N + (sgm - 1) * N : -1;
        const int flags = (sgm < nseg - 1 ? 1 : 0) | (sgm > 0 ? 2 : 0) | (sgm < nseg - 1 ? 4 : 0);
        const int next_off = p0.reverse ? -1 : len;   // the first element (in scan order) of the segment behind this one
        const bool vec = short_vec_bwd(q);
#define VMS_S(Z_, R_)                                                                                                    \
    do {                                                                                                                \
        if (p.seqlen <= 8) {                                                                                            \
            if (vec) hipLaunchKernelGGL((scan_bwd_short_kernel<T, Z_, R_, 8, true>), grid, block, 0, stream, q, ws, gcar, h_in, flags, next_off);         \
            else hipLaunchKernelGGL((scan_bwd_short_kernel<T, Z_, R_, 8, false>), grid, block, 0, stream, q, ws, gcar, h_in, flags, next_off);            \
        } else {                                                                                                        \
            if (vec) hipLaunchKernelGGL((scan_bwd_short_kernel<T, Z_, R_, 16, true>), grid, block, 0, stream, q, ws, gcar, h_in, flags, next_off);        \
            else hipLaunchKernelGGL((scan_bwd_short_kernel<T, Z_, R_, 16, false>), grid, block, 0, stream, q, ws, gcar, h_in, flags, next_off);           \
        }                                                                                                               \
    } while (0)
        if (p.z) { if (p.reverse) VMS_S(true, true); else VMS_S(true, false); }
        else { if (p.reverse) VMS_S(false, true); else VMS_S(false, false); }
#undef VMS_S
        VMS_LAUNCH_CHECK();
    }
    if (ws) {
        const int64_t threads = (int64_t)((p0.batch + kRB - 1) / kRB) * 18 * p0.dim;
        hipLaunchKernelGGL(short_reduce_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, ws, q0);
    }
    VMS_LAUNCH_CHECK();
    set_last_kernel(nseg > 1 ? "scan_bwd_short+segments" : "scan_bwd_short");
    return VMS_OK;
}
int launch_scan_bwd_short(const vms_scan_bwd_params& q, hipStream_t stream) {
    switch (q.f.dtype) {
        case VMS_F32: return launch_bwd_short_t<float>(q, stream);
        case VMS_F16: return launch_bwd_short_t<f16_t>(q, stream);
        default: return launch_bwd_short_t<bf16_t>(q, stream);
    }
}

}  // namespace vms
