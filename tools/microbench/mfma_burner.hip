// tools/mfma_burner.hip -- the matrix-pipe half of the round-3 overlap probe (profiles/r03_overlap.md).
// A persistent-style workgroup of 4 waves (one per SIMD) that issues v_mfma_f32_32x32x16_bf16 back to back on NACC
// independent accumulators, optionally with one ds_read_b128 per MFMA (the LDS rate of a 64x64-per-wave GEMM tile)
// and one 16-byte global load per LOADS_EVERY MFMAs.  It stands in for "a GEMM that was written to fit beside a
// scan workgroup": <= 128 VGPRs, no barrier, LDS request chosen by the caller (dynamic shared memory) so that the
// number of burner workgroups a CU accepts can be pinned from the host.
//   build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/mfma_burner.hip -o tools/build/libmfma_burner.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NACC, bool LDS_READS, bool GLOADS, int PAD = 0>
__global__ __launch_bounds__(256, 1) void burner_kernel(float* __restrict__ out, const f32x4* __restrict__ src, int iters,
                                                        int src_vecs) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    // shader cycles (s_memtime) and wall time (100 MHz constant counter) of this workgroup: effective clock = cycles / wall
    const uint64_t cyc0 = __builtin_readcyclecounter(), wall0 = wall_clock64();
    f32x16 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    union { bf16x8 v; f32x4 f; } a, b;
    a.f = f32x4{1.0f + lane * 1e-3f, 0.5f, 0.25f, 0.125f};
    b.f = f32x4{0.75f, 1.5f - lane * 1e-3f, 0.375f, 0.0625f};
    if (LDS_READS) {
        for (int i = threadIdx.x; i < 4096; i += blockDim.x) smem[i] = 1e-3f * i;
        __syncthreads();
    }
    const f32x4* lsrc = reinterpret_cast<const f32x4*>(smem) + lane;
    uint32_t gidx = (blockIdx.x * 256u + threadIdx.x) % (uint32_t)src_vecs;
    f32x4 g = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) {
            if (LDS_READS) {
                const f32x4 t = lsrc[((it + j) & 15) * 64];
                a.f += t;  // keeps the read live; one v_pk_add pair per MFMA
            }
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc[j], 0, 0, 0);
            if (PAD > 0) __builtin_amdgcn_s_sleep(PAD);   // idle matrix pipe between MFMAs: PAD x 64 clocks
        }
        if (GLOADS) {
            g += src[gidx];
            gidx += 256u * gridDim.x;
            if (gidx >= (uint32_t)src_vecs) gidx -= (uint32_t)src_vecs;
        }
    }
    const uint64_t cyc1 = __builtin_readcyclecounter(), wall1 = wall_clock64();
    if (threadIdx.x == 0) {
        reinterpret_cast<uint64_t*>(out)[2 * blockIdx.x] = cyc1 - cyc0;
        reinterpret_cast<uint64_t*>(out)[2 * blockIdx.x + 1] = wall1 - wall0;
    }
    float s = g.x + g.y + g.z + g.w;
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    if (s == 12345.678f) out[4096 + blockIdx.x * blockDim.x + threadIdx.x] = s;  // never true: keeps the work alive
}

extern "C" {
// variant: 0 = MFMA only, 4 accumulators (saturates the matrix pipe)      1 = + ds_read_b128 (+ v_pk_add) per MFMA
//          2 = + one 16-byte global load per 4 MFMAs too                    3 = MFMA only, ONE accumulator (dependent chain)
//          4 = MFMA only, 4 accumulators, s_sleep 1 (64 clocks) after each  5 = the same with s_sleep 2
// returns MFMA instructions per wave
int burner_launch(int variant, int grid, int iters, int lds_bytes, float* out, const void* src, int src_vecs, hipStream_t stream) {
    typedef void (*kern_t)(float*, const f32x4*, int, int);
    static const kern_t ks[6] = {burner_kernel<4, false, false>, burner_kernel<4, true, false>, burner_kernel<4, true, true>,
                                 burner_kernel<1, false, false>, burner_kernel<4, false, false, 1>, burner_kernel<4, false, false, 2>};
    static const int nacc[6] = {4, 4, 4, 1, 4, 4};
    if (variant < 0 || variant > 5) return -1;
    static bool attr[6] = {false, false, false, false, false, false};
    if (!attr[variant]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ks[variant]), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr[variant] = true;
    }
    if (lds_bytes < 16384) lds_bytes = 16384;
    hipLaunchKernelGGL(ks[variant], dim3(grid), dim3(256), lds_bytes, stream, out, static_cast<const f32x4*>(src), iters, src_vecs);
    return hipGetLastError() == hipSuccess ? nacc[variant] * iters : -1;
}
}
