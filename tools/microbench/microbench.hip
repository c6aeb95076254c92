// microbench.hip -- VALU / transcendental / DPP issue rates on gfx950, to size the scan kernels.
// build: hipcc --offload-arch=gfx950 -O3 tools/microbench/microbench.hip -o tools/build/microbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int NCH = 16;  // independent chains per lane

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float c1, float c2) {
    float x[NCH];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[NCH / 2];
#pragma unroll
    for (int j = 0; j < NCH; ++j) x[j] = threadIdx.x * 1e-3f + j;
#pragma unroll
    for (int j = 0; j < NCH / 2; ++j) p[j] = f2{x[2 * j], x[2 * j + 1]};
    f2 pc1 = {c1, c1}, pc2 = {c2, c2};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(c1), "v"(c2));
            if (MODE == 1 && j < NCH / 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j]) : "v"(pc1), "v"(pc2));
            if (MODE == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(x[j]));
            if (MODE == 3) {  // 1 trans : 4 valu, the forward scan's mix
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[j]) : "v"(c1));
                asm volatile("v_exp_f32 %0, %0" : "+v"(x[j]));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[j]) : "v"(c1));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(c1), "v"(c2));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(c1), "v"(c2));
            }
            if (MODE == 4) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[j]) : "v"(c1));
            if (MODE == 5 && j < NCH / 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[j]) : "v"(pc1));
            if (MODE == 6) asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[j]));
            if (MODE == 7) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[j]));
            if (MODE == 8) {  // 4 valu only (same as mode 3 without the exp)
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[j]) : "v"(c1));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[j]) : "v"(c1));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(c1), "v"(c2));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(c1), "v"(c2));
            }
            if (MODE == 9) asm volatile("v_log_f32 %0, %0" : "+v"(x[j]));
        }
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < NCH; ++j) s += x[j];
#pragma unroll
    for (int j = 0; j < NCH / 2; ++j) s += p[j].x + p[j].y;
    if (s == 12345.678f) out[0] = s;
}

template <int MODE>
int run(const char* name, double ops_per_iter_lane, int waves_per_simd) {
    float* d;
    CHECK(hipMalloc(&d, 4));
    const int iters = 2000;
    dim3 grid(256 * waves_per_simd), block(256);  // 4 waves per block -> waves_per_simd per SIMD
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, d, 10, 0.999f, 0.001f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, d, iters, 0.999f, 0.001f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    double lanes = (double)grid.x * block.x;
    double ops = lanes * iters * ops_per_iter_lane;
    // cycles per wave-instruction per SIMD at 2.4 GHz: 1024 SIMDs
    double winstr = ops / 64.0;
    double cyc = ms * 1e-3 * 2.4e9 * 1024 / winstr;
    printf("%-34s waves/SIMD=%d  %8.3f ms  %8.2f Tlane-op/s  ~%5.2f cyc/wave-instr/SIMD (@2.4GHz)\n", name,
           waves_per_simd, ms, ops / ms * 1e-9, cyc);
    CHECK(hipFree(d));
    return 0;
}

int main() {
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", NCH, w);
        run<1>("v_pk_fma_f32 (2 lanes-ops each)", NCH, w);
        run<4>("v_mul_f32", NCH, w);
        run<5>("v_pk_mul_f32 (2 each)", NCH, w);
        run<2>("v_exp_f32", NCH, w);
        run<9>("v_log_f32", NCH, w);
        run<7>("v_rcp_f32", NCH, w);
        run<3>("mix mul,exp,mul,fma,fma (5 instr)", NCH * 5, w);
        run<8>("mix mul,mul,fma,fma (4 instr)", NCH * 4, w);
        run<6>("v_mov_dpp row_shr:1 (+s_nop 1)", NCH, w);
    }
    return 0;
}
