// Which shader clock does a VALU-bound kernel run at, depending on what the queue held before it?
// (the question behind "the dual backward scan takes 8 % longer in the step than back to back")
// probe: a fixed chain of dependent fp32 fma per lane (VALU-bound like the scans) on every SIMD; workgroup 0 records the shader
// clock (clock64 = s_memtime) and the constant 100 MHz clock (wall_clock64 = s_memrealtime) at its start and end:
//   cycles / time = the effective shader clock of THAT launch; cycles are the same for every launch, time is what moves.
// burn: an MFMA loop of about the length of an in_proj GEMM.
// hipcc --offload-arch=gfx950 -O2 tools/microbench/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <unistd.h>
#include <vector>

__global__ void probe(float* sink, long long* rec, int iters) {
    const long long c0 = clock64(), w0 = wall_clock64();
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 1e-4f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) a = fmaf(a, b, c);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    if (a == 12345.f) sink[0] = a;
    if (blockIdx.x == 0 && threadIdx.x == 0) { rec[0] = c1 - c0; rec[1] = w1 - w0; }
}

typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void burn(float* sink, int iters) {
    s16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + threadIdx.x); b[i] = (short)(0x3f80 + i); }
    f32x4 acc[4] = {};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k], 0, 0, 0);
    }
    if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 12345.f) sink[0] = acc[0][0];
}

int main() {
    float* sink; long long* rec;
    hipMalloc(&sink, 64); hipMalloc(&rec, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * 8, block = 256, probe_iters = 6000, burn_iters = 12000;
    auto run_probe = [&](double& mhz, double& us) {
        hipEventRecord(e0);
        probe<<<grid, block>>>(sink, rec, probe_iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[2]; hipMemcpy(h, rec, sizeof(h), hipMemcpyDeviceToHost);
        mhz = (double)h[0] / ((double)h[1] / 100.0);      // cycles per microsecond
        us = ms * 1e3;
    };
    auto burn_us = [&]() { hipEventRecord(e0); burn<<<grid, block>>>(sink, burn_iters); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3; };
    for (int i = 0; i < 30; ++i) { double m, u; run_probe(m, u); }
    printf("burn kernel: %.0f us\n", burn_us());
    struct Case { const char* name; int mode; } cases[] = {{"back to back", 0}, {"after two MFMA burns", 1}, {"after 2 ms of idle queue", 2}, {"back to back again", 0}};
    for (auto& c : cases) {
        std::vector<double> mhz, us;
        for (int i = 0; i < 25; ++i) {
            if (c.mode == 1) { burn<<<grid, block>>>(sink, burn_iters); burn<<<grid, block>>>(sink, burn_iters); }
            if (c.mode == 2) { hipDeviceSynchronize(); usleep(2000); }
            double m, u; run_probe(m, u);
            if (i >= 5) { mhz.push_back(m); us.push_back(u); }
        }
        std::sort(mhz.begin(), mhz.end()); std::sort(us.begin(), us.end());
        printf("%-26s probe %7.1f us (min %7.1f)   workgroup 0: shader clock %6.0f MHz (min %6.0f, max %6.0f)\n", c.name, us[us.size() / 2], us[0],
               mhz[mhz.size() / 2], mhz[0], mhz.back());
    }
    return 0;
}
