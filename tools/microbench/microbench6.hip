// microbench6.hip -- latency of DEPENDENT VALU chains on gfx950 at 1 and 2 waves per SIMD:
// cycles per instruction of one wave when every instruction depends on the one ILP positions earlier.
// build: hipcc --offload-arch=gfx950 -O3 tools/microbench/microbench6.hip -o tools/build/microbench6
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE, int ILP>
__global__ __launch_bounds__(256) void k(float* out, int iters, float c1, float c2) {
    float x[ILP];
    f2 p[ILP];
#pragma unroll
    for (int j = 0; j < ILP; ++j) { x[j] = threadIdx.x * 1e-3f + j; p[j] = f2{x[j], x[j] + 1.f}; }
    f2 pc1 = {c1, c1}, pc2 = {c2, c2};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 16 / ILP; ++rep) {
#pragma unroll
            for (int j = 0; j < ILP; ++j) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(c1), "v"(c2));
                if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j]) : "v"(pc1), "v"(pc2));
                if (MODE == 2) asm volatile("v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(x[j]) : "v"(c1), "v"(c2));
                if (MODE == 3) asm volatile("v_exp_f32 %0, %0\n\tv_mul_f32 %0, %0, %1" : "+v"(x[j]) : "v"(c1));
                if (MODE == 4) asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[j]) : "v"(c1));
                if (MODE == 5) asm volatile("v_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[j]) : "v"(c1));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < ILP; ++j) s += x[j] + p[j].x + p[j].y;
    if (s == 12345.678f) out[0] = s;
}

template <int MODE, int ILP>
int run(const char* name, double instr_per_slot, int waves_per_simd) {
    float* d;
    CHECK(hipMalloc(&d, 4));
    const int iters = 4000;
    dim3 grid(256 * waves_per_simd), block(256);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<MODE, ILP>), grid, block, 0, 0, d, 2000, 0.999f, 0.001f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<MODE, ILP>), grid, block, 0, 0, d, iters, 0.999f, 0.001f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_wave = (double)iters * 16 * instr_per_slot;
    printf("%-44s ILP=%2d waves/SIMD=%d  %8.3f ms  %6.2f cycles per instruction of a wave (@2.4 GHz)\n", name, ILP,
           waves_per_simd, ms, ms * 1e-3 * 2.4e9 / instr_per_wave);
    CHECK(hipFree(d));
    return 0;
}

#define ALL_ILP(M, NAME, IPS, W) run<M, 1>(NAME, IPS, W); run<M, 2>(NAME, IPS, W); run<M, 4>(NAME, IPS, W); run<M, 8>(NAME, IPS, W); run<M, 16>(NAME, IPS, W)
int main() {
    for (int w = 1; w <= 2; ++w) {
        ALL_ILP(0, "v_fma_f32 chain", 1, w);
        ALL_ILP(1, "v_pk_fma_f32 chain", 1, w);
        ALL_ILP(2, "v_mul_f32 -> v_add_f32 chain", 2, w);
        ALL_ILP(3, "v_exp_f32 -> v_mul_f32 chain", 2, w);
        ALL_ILP(4, "s_nop 1; v_fmac_f32_dpp row_shr:1 chain", 1, w);
    }
    return 0;
}
