// tr_probe.hip -- semantics of ds_read_b64_tr_b16 on gfx950: which LDS elements does lane l receive when lane i of a 16-lane
// group supplies the address of row i / 4, column group i % 4 of a [4][16] matrix of 16-bit elements with a free row stride?
// hipcc --offload-arch=gfx950 -O2 tools/microbench/tr_probe.hip -o tools/build/tr_probe && tools/build/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(uint16_t* out, int row_stride_elems, int group_stride_elems) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x, i = l & 15, grp = l >> 4;
    const uint32_t addr = (uint32_t)(uintptr_t)(lds) + 2u * (grp * group_stride_elems + (i >> 2) * row_stride_elems + (i & 3) * 4);
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[4 * l + 0] = v.x & 0xffff; out[4 * l + 1] = v.x >> 16; out[4 * l + 2] = v.y & 0xffff; out[4 * l + 3] = v.y >> 16;
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    const int cfg[3][2] = {{16, 64}, {64, 16}, {128, 1024}};
    for (auto& c : cfg) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, c[0], c[1]);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("row stride %d elems, 16-lane group stride %d elems\n", c[0], c[1]);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d: %5d %5d %5d %5d", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
            if (l % 2) printf("\n");
        }
    }
    return 0;
}
