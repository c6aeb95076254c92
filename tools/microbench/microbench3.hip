// microbench3.hip -- ceiling of the "lane = row, states in registers, B/C in SGPRs" scan loop on gfx950.
// Each wave = 64 rows x T elements x 16 states, sequential over T; B/C are wave-uniform (scalar loads).
// build: hipcc --offload-arch=gfx950 -O3 tools/microbench/microbench3.hip -o tools/build/microbench3
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int N = 16;

__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// MODE 0: pass 2 (mul, exp, mul, fma, fma)   MODE 1: pass 1 (mul, exp, mul, fma)
// BCF32: B/C given as fp32 (no SALU conversion)
template <int MODE, bool BCF32, int WPS>
__global__ __launch_bounds__(256, WPS) void k(const float* __restrict__ dl, const float* __restrict__ du,
                                               const void* __restrict__ Bp, const void* __restrict__ Cp,
                                               const float* __restrict__ Ap, float* __restrict__ out, int T, int L) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wid = blockIdx.x * 4 + wave;  // wave id
    const int nchunk = L / T;
    const int rb = wid / nchunk, c = wid % nchunk;  // row block, chunk
    float A[N], x[N];
#pragma unroll
    for (int n = 0; n < N; ++n) { A[n] = Ap[(rb * 64 + lane) * N + n]; x[n] = 0.f; }
    // dl/du laid out [rb][l][64 lanes] fp32 (as if already transposed): coalesced
    const float* dlp = dl + ((size_t)rb * L + (size_t)c * T) * 64 + lane;
    const float* dup = du + ((size_t)rb * L + (size_t)c * T) * 64 + lane;
    float* op = out + ((size_t)rb * L + (size_t)c * T) * 64 + lane;
    const int l0 = c * T;
    for (int l = 0; l < T; l += 8) {
        float d[8], v[8], y[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { d[i] = dlp[(l + i) * 64]; v[i] = dup[(l + i) * 64]; y[i] = 0.f; }
#pragma unroll
        for (int n = 0; n < N; ++n) {
            float Bs[8], Cs[8];
            if (BCF32) {
                const float* Bf = (const float*)Bp + (size_t)n * L + l0 + l;
                const float* Cf = (const float*)Cp + (size_t)n * L + l0 + l;
#pragma unroll
                for (int i = 0; i < 8; ++i) { Bs[i] = Bf[i]; Cs[i] = Cf[i]; }
            } else {
                const uint32_t* Bw = (const uint32_t*)((const uint16_t*)Bp + (size_t)n * L + l0 + l);
                const uint32_t* Cw = (const uint32_t*)((const uint16_t*)Cp + (size_t)n * L + l0 + l);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    uint32_t wb = Bw[i], wc = Cw[i];
                    Bs[2 * i] = bf_lo(wb); Bs[2 * i + 1] = bf_hi(wb);
                    Cs[2 * i] = bf_lo(wc); Cs[2 * i + 1] = bf_hi(wc);
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float a = __builtin_amdgcn_exp2f(d[i] * A[n]);
                x[n] = fmaf(a, x[n], v[i] * Bs[i]);
                if (MODE == 0) y[i] = fmaf(Cs[i], x[n], y[i]);
            }
        }
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) op[(l + i) * 64] = y[i];
        }
    }
    if (MODE == 1) {
        float s = 0.f;
#pragma unroll
        for (int n = 0; n < N; ++n) s += x[n];
        op[0] = s;
    }
}

template <int MODE, bool BCF32, int WPS>
int run(const char* name, int T) {
    const int rows = 8192, L = 8192, RB = rows / 64;
    float *dl, *du, *A, *out;
    void *B, *C;
    size_t ne = (size_t)rows * L;
    CHECK(hipMalloc(&dl, ne * 4)); CHECK(hipMalloc(&du, ne * 4)); CHECK(hipMalloc(&out, ne * 4));
    CHECK(hipMalloc(&A, rows * N * 4));
    CHECK(hipMalloc(&B, (size_t)N * L * 4)); CHECK(hipMalloc(&C, (size_t)N * L * 4));
    std::vector<float> h(ne);
    for (size_t i = 0; i < ne; ++i) h[i] = 0.001f + 0.01f * ((i * 2654435761u) % 1000) / 1000.f;
    CHECK(hipMemcpy(dl, h.data(), ne * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(du, h.data(), ne * 4, hipMemcpyHostToDevice));
    std::vector<float> ha(rows * N);
    for (int i = 0; i < rows * N; ++i) ha[i] = -(1 + i % 16) * 1.44f;
    CHECK(hipMemcpy(A, ha.data(), rows * N * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(B, 0x3f, (size_t)N * L * 4)); CHECK(hipMemset(C, 0x3f, (size_t)N * L * 4));
    const int waves = RB * (L / T);
    dim3 grid(waves / 4), block(256);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<MODE, BCF32, WPS>), grid, block, 0, 0, dl, du, B, C, A, out, T, L);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<MODE, BCF32, WPS>), grid, block, 0, 0, dl, du, B, C, A, out, T, L);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    double es = (double)ne * N;
    printf("%-40s T=%4d waves=%5d  %8.1f us   %6.2f cyc/(wave elem-state)/SIMD@2.4GHz   %6.1f G elem-state/s\n", name, T, waves,
           ms * 1e3, ms * 1e-3 * 2.4e9 * 1024 / (es / 64), es / ms * 1e-6);
    hipFree(dl); hipFree(du); hipFree(out); hipFree(A); hipFree(B); hipFree(C);
    return 0;
}

int main() {
    for (int T : {128, 256, 512, 1024}) {
        run<0, false, 8>("pass2 bf16 B/C, 8 waves/SIMD bound", T);
        run<0, true, 8>("pass2 fp32 B/C, 8 waves/SIMD bound", T);
        run<1, false, 8>("pass1 bf16 B/C, 8 waves/SIMD bound", T);
        run<1, true, 8>("pass1 fp32 B/C", T);
    }
    run<0, false, 4>("pass2 bf16 B/C, 4 waves/SIMD bound", 256);
    run<0, true, 4>("pass2 fp32 B/C, 4 waves/SIMD bound", 256);
    return 0;
}
