// microbench2.hip -- LDS op rates (ds_add_f32 vs ds_write/ds_read), global fp32 atomics, v_exp_f16.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef __attribute__((address_space(3))) float lds_f32;
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float* gbuf, int iters) {
    __shared__ float sm[256 * 16 + 64];
    lds_f32* s = (lds_f32*)sm;
    const int t = threadIdx.x;
    for (int i = t; i < 256 * 16; i += 256) sm[i] = 0.f;
    __syncthreads();
    float acc = 0.f;
    f4 v4 = {1.f, 2.f, 3.f, 4.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (MODE == 0) __hip_atomic_fetch_add(s + j * 256 + t, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 1) acc += __hip_atomic_fetch_add(s + j * 256 + t, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 2) { s[j * 256 + t] = acc + j; asm volatile("" ::: "memory"); }
            if (MODE == 3 && j < 4) { *(__attribute__((address_space(3))) f4*)(s + (j * 256 + t) * 4) = v4; asm volatile("" ::: "memory"); }
            if (MODE == 4 && j < 4) { f4 r = *(__attribute__((address_space(3))) f4*)(s + (j * 256 + t) * 4); acc += r.x + r.y + r.z + r.w; asm volatile("" ::: "memory"); }
            if (MODE == 5) { acc += s[j * 256 + t]; asm volatile("" ::: "memory"); }
            if (MODE == 6) __hip_atomic_fetch_add(gbuf + ((size_t)blockIdx.x * 16 + j) * 256 + t, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (MODE == 7) { _Float16 h = (_Float16)acc; asm volatile("v_exp_f16 %0, %0" : "+v"(h)); acc = (float)h; }
            if (MODE == 8) __hip_atomic_fetch_add(gbuf + (size_t)(j * 256 + t), 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // all blocks same 16 KB
        }
    }
    __syncthreads();
    if (acc == 1234.5f || sm[t] == 1234.5f) out[0] = acc;
}

template <int MODE>
int run(const char* name, double bytes_per_op, double ops_per_iter_lane, int wgs_per_cu, int iters) {
    float *d, *g;
    CHECK(hipMalloc(&d, 4));
    dim3 grid(256 * wgs_per_cu), block(256);
    CHECK(hipMalloc(&g, (size_t)grid.x * 16 * 256 * 4));
    CHECK(hipMemset(g, 0, (size_t)grid.x * 16 * 256 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, d, g, 2);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, d, g, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double ops = (double)grid.x * block.x * iters * ops_per_iter_lane;
    double per_cu_clk = ops / 256.0 / (ms * 1e-3 * 2.4e9);
    printf("%-40s WG/CU=%d %8.3f ms  %9.2f Glane-op/s  %6.2f lane-op/clk/CU  %7.1f B/clk/CU\n", name, wgs_per_cu, ms,
           ops / ms * 1e-6, per_cu_clk, per_cu_clk * bytes_per_op);
    CHECK(hipFree(d)); CHECK(hipFree(g));
    return 0;
}

int main() {
    for (int w : {2, 4}) {
        run<0>("ds_add_f32 (no return), conflict-free", 4, 16, w, 500);
        run<1>("ds_add_rtn_f32", 4, 16, w, 500);
        run<2>("ds_write_b32", 4, 16, w, 500);
        run<3>("ds_write_b128", 16, 4, w, 500);
        run<4>("ds_read_b128", 16, 4, w, 500);
        run<5>("ds_read_b32", 4, 16, w, 500);
        run<6>("global_atomic_add_f32 coalesced, disjoint", 4, 16, w, 50);
        run<8>("global_atomic_add_f32 coalesced, same 16KB", 4, 16, w, 50);
        run<7>("v_exp_f16 (+2 cvt)", 4, 16, w, 500);
    }
    return 0;
}
