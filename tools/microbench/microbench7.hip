// microbench7: cost of v_permlane32_swap / v_permlane16_swap / DPP adds / ds_bpermute next to packed math (gfx950)
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench/microbench7.hip -o tools/build/microbench7 ; run: tools/build/microbench7
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int V>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    float a = threadIdx.x * 0.5f, b = threadIdx.x * 0.25f, c = 1.f, d = 2.f, e = 3.f, f = 4.f, g = 5.f, h = 6.f;
    for (int it = 0; it < iters; ++it) {
        if (V == 0) asm volatile(REP8("v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\t")
                                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
        if (V == 1) asm volatile(REP8("v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\tv_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7\n\t")
                                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
        if (V == 2) asm volatile(REP8("v_add_f32_dpp %0, %1, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %2, %3, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                                      "v_add_f32_dpp %4, %5, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %6, %7, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n\t")
                                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
        if (V == 3) asm volatile(REP8("v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3\n\tv_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3\n\t")
                                 : "+v"(*(double*)&a), "+v"(*(double*)&c) : "v"(*(double*)&e), "v"(*(double*)&g));
        if (V == 4) asm volatile(REP8("v_add_f32 %0, %0, %1\n\tv_add_f32 %2, %2, %3\n\tv_add_f32 %4, %4, %5\n\tv_add_f32 %6, %6, %7\n\t")
                                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
        if (V == 5) asm volatile(REP8("v_mov_b32_dpp %0, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\tv_mov_b32_dpp %2, %3 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                                      "v_mov_b32_dpp %4, %5 row_bcast:15 row_mask:0xa bank_mask:0xf\n\tv_mov_b32_dpp %6, %7 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t")
                                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h;
}
template <int V>
void run(const char* name, float* d, int wgs) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(wgs), dim3(512), 0, 0, d, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(wgs), dim3(512), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = wgs * 8.0 / 1024.0;
    printf("%-34s waves/SIMD=%.0f  %8.3f ms  %6.2f cycles per instruction per SIMD (@2.4 GHz nominal)\n", name, waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / (iters * 32.0 * waves_per_simd));
}
int main() {
    float* d; hipMalloc(&d, 1024 * 512 * 4);
    for (int wgs : {256, 512}) {
        run<0>("v_permlane32_swap_b32", d, wgs); run<1>("v_permlane16_swap_b32", d, wgs); run<2>("v_add_f32_dpp row_ror:8", d, wgs);
        run<3>("v_pk_add_f32", d, wgs); run<4>("v_add_f32", d, wgs); run<5>("v_mov_b32_dpp row_bcast", d, wgs);
    }
    return 0;
}
