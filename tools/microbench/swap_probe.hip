// v_permlane32_swap / v_permlane16_swap: the inline-asm form vs the compiler builtins (operand and result order).
// hipcc --offload-arch=gfx950 -O2 tools/microbench/swap_probe.hip -o /tmp/swap_probe && /tmp/swap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    const int lane = threadIdx.x;
    int a = lane, b = 100 + lane;          // a = "vdst", b = "src0"
    int a1 = a, b1 = b;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a1), "+v"(b1));
    auto r = __builtin_amdgcn_permlane32_swap((unsigned)a, (unsigned)b, false, false);
    int a2 = a, b2 = b;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a2), "+v"(b2));
    auto q = __builtin_amdgcn_permlane16_swap((unsigned)a, (unsigned)b, false, false);
    out[lane] = a1; out[64 + lane] = b1; out[128 + lane] = (int)r[0]; out[192 + lane] = (int)r[1];
    out[256 + lane] = a2; out[320 + lane] = b2; out[384 + lane] = (int)q[0]; out[448 + lane] = (int)q[1];
}
int main() {
    int* d; hipMalloc(&d, 512 * 4); k<<<1, 64>>>(d); int h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[8] = {"asm32 a", "asm32 b", "blt32 r0", "blt32 r1", "asm16 a", "asm16 b", "blt16 r0", "blt16 r1"};
    for (int r = 0; r < 8; ++r) { printf("%-9s", names[r]); for (int i = 0; i < 64; i += 8) printf(" %3d", h[64 * r + i]); printf("\n"); }
    int same32 = 1, same16 = 1;
    for (int i = 0; i < 64; ++i) { same32 &= h[i] == h[128 + i] && h[64 + i] == h[192 + i]; same16 &= h[256 + i] == h[384 + i] && h[320 + i] == h[448 + i]; }
    printf("builtin == asm: permlane32 %d, permlane16 %d\n", same32, same16);
}
