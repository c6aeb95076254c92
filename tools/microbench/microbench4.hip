// microbench4.hip -- "lane = row" scan loop, two ways of feeding the wave-uniform B/C values:
//   S: scalar loads (s_load_dwordx8, fp32 B/C) software-prefetched one state ahead (inline asm)
//   L: fp32 B/C tile staged in LDS once per workgroup, broadcast ds_read_b128
// build: hipcc --offload-arch=gfx950 -O3 tools/microbench/microbench4.hip -o tools/build/microbench4
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int N = 16;
typedef float f8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define SLOAD8(dst, base, off) asm volatile("s_load_dwordx8 %0, %1, %2" : "=&s"(dst) : "s"(base), "s"(off))
#define SWAIT2(a, b) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b))
#define SWAIT2D(a, b, dep) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b), "+v"(dep))

// ---------------- S: scalar-load fed ----------------
__device__ unsigned long long g_clk[2];
template <int MODE, int WPS, bool NOMEM = false>
__global__ __launch_bounds__(256, WPS) void kS(const float* __restrict__ dl, const float* __restrict__ du,
                                                const float* __restrict__ Bp, const float* __restrict__ Cp,
                                                const float* __restrict__ Ap, float* __restrict__ out, int T, int L) {
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wid = blockIdx.x * 4 + wave;
    const int nchunk = L / T;
    const int rb = wid / nchunk, c = wid % nchunk;
    float A[N], x[N];
#pragma unroll
    for (int n = 0; n < N; ++n) { A[n] = Ap[(rb * 64 + lane) * N + n]; x[n] = 0.f; }
    const float* dlp = dl + ((size_t)rb * L + (size_t)c * T) * 64 + lane;
    const float* dup = du + ((size_t)rb * L + (size_t)c * T) * 64 + lane;
    float* op = out + ((size_t)rb * L + (size_t)c * T) * 64 + lane;
    const float* Bb = Bp + c * T;
    const float* Cb = Cp + c * T;
    uint32_t off[N];
#pragma unroll
    for (int n = 0; n < N; ++n) off[n] = __builtin_amdgcn_readfirstlane(n * L * 4);
    f8 Bq0, Cq0, Bq1, Cq1;
    SLOAD8(Bq0, Bb, off[0]);
    SLOAD8(Cq0, Cb, off[0]);
    for (int l = 0; l < T; l += 8) {
        float d[8], v[8], y[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (NOMEM) { d[i] = 0.001f * (l + i + lane); v[i] = 0.002f * (l + i + lane); }
            else { d[i] = dlp[(l + i) * 64]; v[i] = dup[(l + i) * 64]; }
            y[i] = 0.f;
        }
        const float* Bn = Bb + 8;  // next step
        const float* Cn = Cb + 8;
#pragma unroll
        for (int n = 0; n < N; n += 2) {
            if (n == 0) SWAIT2(Bq0, Cq0); else SWAIT2D(Bq0, Cq0, x[n - 1]);
            SLOAD8(Bq1, Bb, off[n + 1]);
            SLOAD8(Cq1, Cb, off[n + 1]);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float a = __builtin_amdgcn_exp2f(d[i] * A[n]);
                x[n] = fmaf(a, x[n], v[i] * Bq0[i]);
                if (MODE == 0) y[i] = fmaf(Cq0[i], x[n], y[i]);
            }
            SWAIT2D(Bq1, Cq1, x[n]);
            if (n + 2 < N) { SLOAD8(Bq0, Bb, off[n + 2]); SLOAD8(Cq0, Cb, off[n + 2]); }
            else { SLOAD8(Bq0, Bn, off[0]); SLOAD8(Cq0, Cn, off[0]); }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float a = __builtin_amdgcn_exp2f(d[i] * A[n + 1]);
                x[n + 1] = fmaf(a, x[n + 1], v[i] * Bq1[i]);
                if (MODE == 0) y[i] = fmaf(Cq1[i], x[n + 1], y[i]);
            }
        }
        Bb = Bn; Cb = Cn;
        if (MODE == 0) {
            if (NOMEM) { float s = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) s += y[i];
                x[0] += s * 1e-30f;
            } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) op[(l + i) * 64] = y[i];
            }
        }
    }
    SWAIT2(Bq0, Cq0);
    if (NOMEM) { float s = 0.f;
#pragma unroll
        for (int n = 0; n < N; ++n) s += x[n];
        op[0] = s; }
    if (threadIdx.x == 0 && blockIdx.x == 7) { g_clk[0] = __builtin_readcyclecounter() - t0; g_clk[1] = wall_clock64() - w0; }
    if (MODE == 1) {
        float s = 0.f;
#pragma unroll
        for (int n = 0; n < N; ++n) s += x[n];
        op[0] = s;
    }
}

// ---------------- L: LDS-broadcast fed ----------------
// workgroup = 4 waves = 4 row blocks of the same chunk; LDS tile [T/E][N][2][E] fp32 (E elements per step)
template <int MODE, int WPS, int E>
__global__ __launch_bounds__(256, WPS) void kL(const float* __restrict__ dl, const float* __restrict__ du,
                                                const float* __restrict__ Bp, const float* __restrict__ Cp,
                                                const float* __restrict__ Ap, float* __restrict__ out, int T, int L) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nchunk = L / T;
    const int c = blockIdx.x % nchunk;
    const int rb = (blockIdx.x / nchunk) * 4 + wave;
    // stage: sm[((l/E)*N + n)*2*E + {0: B, E: C} + l%E]
    for (int j = threadIdx.x; j < N * T; j += 256) {
        const int n = j / T, l = j % T;
        float* dst = sm + ((l / E) * N + n) * 2 * E + (l % E);
        dst[0] = Bp[(size_t)n * L + c * T + l];
        dst[E] = Cp[(size_t)n * L + c * T + l];
    }
    float A[N], x[N];
#pragma unroll
    for (int n = 0; n < N; ++n) { A[n] = Ap[(rb * 64 + lane) * N + n]; x[n] = 0.f; }
    const float* dlp = dl + ((size_t)rb * L + (size_t)c * T) * 64 + lane;
    const float* dup = du + ((size_t)rb * L + (size_t)c * T) * 64 + lane;
    float* op = out + ((size_t)rb * L + (size_t)c * T) * 64 + lane;
    __syncthreads();
    for (int l = 0; l < T; l += E) {
        float d[E], v[E], y[E];
#pragma unroll
        for (int i = 0; i < E; ++i) { d[i] = dlp[(l + i) * 64]; v[i] = dup[(l + i) * 64]; y[i] = 0.f; }
        const f4* row = reinterpret_cast<const f4*>(sm + (l / E) * N * 2 * E);
#pragma unroll
        for (int n = 0; n < N; ++n) {
            float Bs[E], Cs[E];
#pragma unroll
            for (int q = 0; q < E / 4; ++q) {
                f4 b = row[n * (2 * E / 4) + q], cc = row[n * (2 * E / 4) + E / 4 + q];
#pragma unroll
                for (int e = 0; e < 4; ++e) { Bs[4 * q + e] = b[e]; Cs[4 * q + e] = cc[e]; }
            }
#pragma unroll
            for (int i = 0; i < E; ++i) {
                float a = __builtin_amdgcn_exp2f(d[i] * A[n]);
                x[n] = fmaf(a, x[n], v[i] * Bs[i]);
                if (MODE == 0) y[i] = fmaf(Cs[i], x[n], y[i]);
            }
        }
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < E; ++i) op[(l + i) * 64] = y[i];
        }
    }
    if (MODE == 1) {
        float s = 0.f;
#pragma unroll
        for (int n = 0; n < N; ++n) s += x[n];
        op[0] = s;
    }
}

struct Bufs { float *dl, *du, *A, *out, *B, *C; };
static Bufs g;
constexpr int ROWS = 8192, LL = 8192;

template <typename F>
int timeit(const char* name, int T, int waves, F launch) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipGetLastError());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    double es = (double)ROWS * LL * N;
    unsigned long long hc[2] = {0, 0};
    CHECK(hipMemcpyFromSymbol(hc, HIP_SYMBOL(g_clk), 16));
    printf("%-44s T=%4d waves=%5d  %8.1f us   %6.2f cyc/(wave elem-state)/SIMD@2.4GHz   clk %.0f MHz\n", name, T, waves, ms * 1e3,
           ms * 1e-3 * 2.4e9 * 1024 / (es / 64), hc[1] ? 100.0 * hc[0] / hc[1] : 0.0);
    return 0;
}

template <int MODE, int WPS, bool NOMEM = false>
int runS(const char* name, int T) {
    const int waves = ROWS / 64 * (LL / T);
    return timeit(name, T, waves, [&] { hipLaunchKernelGGL((kS<MODE, WPS, NOMEM>), dim3(waves / 4), dim3(256), 0, 0, g.dl, g.du, g.B, g.C, g.A, g.out, T, LL); });
}
template <int MODE, int WPS, int E>
int runL(const char* name, int T) {
    const int waves = ROWS / 64 * (LL / T);
    return timeit(name, T, waves, [&] { hipLaunchKernelGGL((kL<MODE, WPS, E>), dim3(waves / 4), dim3(256), T * N * 2 * 4, 0, g.dl, g.du, g.B, g.C, g.A, g.out, T, LL); });
}

int main() {
    size_t ne = (size_t)ROWS * LL;
    CHECK(hipMalloc(&g.dl, ne * 4)); CHECK(hipMalloc(&g.du, ne * 4)); CHECK(hipMalloc(&g.out, ne * 4));
    CHECK(hipMalloc(&g.A, ROWS * N * 4));
    CHECK(hipMalloc(&g.B, (size_t)N * LL * 4 + 256)); CHECK(hipMalloc(&g.C, (size_t)N * LL * 4 + 256));
    std::vector<float> h(ne);
    for (size_t i = 0; i < ne; ++i) h[i] = 0.001f + 0.01f * ((i * 2654435761u) % 1000) / 1000.f;
    CHECK(hipMemcpy(g.dl, h.data(), ne * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(g.du, h.data(), ne * 4, hipMemcpyHostToDevice));
    std::vector<float> ha(ROWS * N);
    for (int i = 0; i < ROWS * N; ++i) ha[i] = -(1 + i % 16) * 1.44f;
    CHECK(hipMemcpy(g.A, ha.data(), ROWS * N * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(g.B, 0x3f, (size_t)N * LL * 4 + 256)); CHECK(hipMemset(g.C, 0x3f, (size_t)N * LL * 4 + 256));
    for (int T : {128, 256}) {
        runS<0, 8, true>("S pass2 NOMEM (8 w/SIMD bound)", T);
        runS<0, 4, true>("S pass2 NOMEM (4 w/SIMD bound)", T);
        runS<1, 8, true>("S pass1 NOMEM (8 w/SIMD bound)", T);
        runS<0, 8>("S pass2 (8 w/SIMD bound)", T);
        runS<0, 4>("S pass2 (4 w/SIMD bound)", T);
        runS<1, 8>("S pass1 (8 w/SIMD bound)", T);
        runL<0, 8, 4>("L pass2 E=4 (8 w/SIMD bound)", T);
        runL<0, 4, 4>("L pass2 E=4 (4 w/SIMD bound)", T);
        runL<0, 4, 8>("L pass2 E=8 (4 w/SIMD bound)", T);
        runL<1, 8, 4>("L pass1 E=4 (8 w/SIMD bound)", T);
        runL<1, 4, 8>("L pass1 E=8 (4 w/SIMD bound)", T);
    }
    return 0;
}
