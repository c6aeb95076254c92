// probes the lane-wise semantics of v_pk_mul_f32 / v_pk_fma_f32 operand forms on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(float* out, float s0, float s1) {
    const int lane = threadIdx.x;
    f2 a = {1.f + lane, 100.f + lane};
    f2 v = {2.f, 3.f};
    f2 sp = {s0, s1};
    f2 r0, r1, r2, r3, r4;
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r0) : "v"(a), "v"(v));                 // (a.x*2, a.x*3)
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r1) : "v"(a), "v"(v));   // (a.y*2, a.y*3)
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r2) : "v"(a), "s"(sp));                // (a.x*s0, a.x*s1)
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r3) : "v"(a), "s"(sp));  // (a.y*s0, a.y*s1)
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r4) : "s"(sp), "v"(a), "v"(v));                     // (s0*a.x+2, s1*a.y+3)
    float* o = out + lane * 10;
    o[0] = r0.x; o[1] = r0.y; o[2] = r1.x; o[3] = r1.y; o[4] = r2.x; o[5] = r2.y; o[6] = r3.x; o[7] = r3.y; o[8] = r4.x; o[9] = r4.y;
}
int main() {
    float* d; hipMalloc(&d, 64 * 10 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 5.f, 7.f);
    float h[640]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad[5] = {0, 0, 0, 0, 0};
    for (int l = 0; l < 64; ++l) {
        float ax = 1.f + l, ay = 100.f + l;
        float e[10] = {ax * 2, ax * 3, ay * 2, ay * 3, ax * 5, ax * 7, ay * 5, ay * 7, 5 * ax + 2, 7 * ay + 3};
        for (int j = 0; j < 10; ++j) if (h[l * 10 + j] != e[j]) { bad[j / 2]++; if (l < 8) printf("lane %d out %d: got %g want %g\n", l, j, h[l * 10 + j], e[j]); }
    }
    printf("mismatching lanes x2: vv_b0 %d  vv_b1 %d  vs_b0 %d  vs_b1 %d  fma_s %d\n", bad[0], bad[1], bad[2], bad[3], bad[4]);
    return 0;
}
