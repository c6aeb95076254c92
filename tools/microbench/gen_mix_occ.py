#!/usr/bin/env python3
"""Generates tools/build/mix_occ.hip: VALU issue-cost probes for gfx950 at a CONTROLLED number of waves per SIMD.

Every body is one asm block of explicit registers (no compiler scheduling), looped `iters` times by 256-thread workgroups
(one wave per SIMD); dynamic LDS limits how many workgroups a CU holds = waves per SIMD (1, 2, 3, 4, 8).  Output per body and
occupancy: real (clock-corrected) cycles of one SIMD per instruction.  Questions:
  * what a VGPR bank conflict costs (bank = index mod 4) for 2- / 3-source scalar fp32 and for packed fp32;
  * what the backward scan's instruction mix costs per instruction at 2 vs 4 waves per SIMD, with independent work and with
    the kernel's dependent chains.
usage: python tools/microbench/gen_mix_occ.py > tools/build/mix_occ.hip; hipcc --offload-arch=gfx950 -O3 tools/build/mix_occ.hip -o tools/build/mix_occ
"""
import re

bodies = []


def body(name, ins):
    bodies.append((name, ins))


def rep(n, f):
    out = []
    for i in range(n): out += f(i)
    return out


# accumulators v[64 ...], operands v[8 ...]
body("fma 3 src distinct banks", rep(32, lambda i: [f"v_fma_f32 v{64 + i}, v8, v9, v10"]))
body("fma 2 of 3 src same bank", rep(32, lambda i: [f"v_fma_f32 v{64 + i}, v8, v12, v9"]))
body("fma 3 src same bank", rep(32, lambda i: [f"v_fma_f32 v{64 + i}, v8, v12, v16"]))
body("fmac acc (dst = src2), banks distinct", rep(32, lambda i: [f"v_fmac_f32 v{64 + i}, v{8 + (i + 1) % 4}, v{12 + (i + 2) % 4}"]))
body("fmac acc, src0/src1 same bank", rep(32, lambda i: [f"v_fmac_f32 v{64 + i}, v{8 + (i + 1) % 4}, v{12 + (i + 1) % 4}"]))
body("mul 2 src distinct banks", rep(32, lambda i: [f"v_mul_f32 v{64 + i}, v8, v9"]))
body("mul 2 src same bank", rep(32, lambda i: [f"v_mul_f32 v{64 + i}, v8, v12"]))
body("mul sgpr x vgpr", rep(32, lambda i: [f"v_mul_f32 v{64 + i}, s20, v8"]))
body("pk_fma banks {01}{23}{01}", rep(16, lambda i: [f"v_pk_fma_f32 v[{64 + 2 * i}:{65 + 2 * i}], v[8:9], v[10:11], v[12:13]"]))
body("pk_fma banks {01}{01}{01}", rep(16, lambda i: [f"v_pk_fma_f32 v[{64 + 2 * i}:{65 + 2 * i}], v[8:9], v[12:13], v[16:17]"]))
body("pk_fma acc dst = src2", rep(16, lambda i: [f"v_pk_fma_f32 v[{64 + 2 * i}:{65 + 2 * i}], v[8:9], v[10:11], v[{64 + 2 * i}:{65 + 2 * i}]"]))
body("pk_fma sgpr pair operand", rep(16, lambda i: [f"v_pk_fma_f32 v[{64 + 2 * i}:{65 + 2 * i}], v[8:9], s[20:21], v[10:11]"]))
body("pk_mul banks {01}{23}", rep(16, lambda i: [f"v_pk_mul_f32 v[{64 + 2 * i}:{65 + 2 * i}], v[8:9], v[10:11]"]))
body("pk_mul banks {01}{01}", rep(16, lambda i: [f"v_pk_mul_f32 v[{64 + 2 * i}:{65 + 2 * i}], v[8:9], v[12:13]"]))
body("pk_add banks {01}{23}", rep(16, lambda i: [f"v_pk_add_f32 v[{64 + 2 * i}:{65 + 2 * i}], v[8:9], v[10:11]"]))
body("exp", rep(16, lambda i: [f"v_exp_f32 v{64 + i}, v{8 + i % 8}"]))
body("exp, fma, fma, fma (distinct banks)", rep(8, lambda i: [f"v_exp_f32 v{64 + i}, v{8 + i}", f"v_fma_f32 v{72 + i}, v8, v9, v10", f"v_fma_f32 v{80 + i}, v9, v10, v11",
                                                    f"v_fma_f32 v{88 + i}, v8, v9, v10"]))
body("exp, pk_fma, pk_fma", rep(8, lambda i: [f"v_exp_f32 v{64 + i}, v{8 + i}", f"v_pk_fma_f32 v[{72 + 2 * i}:{73 + 2 * i}], v[8:9], v[10:11], v[12:13]",
                                          f"v_pk_fma_f32 v[{88 + 2 * i}:{89 + 2 * i}], v[8:9], v[10:11], v[12:13]"]))
body("8 exp then 24 fma (batched)", rep(8, lambda i: [f"v_exp_f32 v{64 + i}, v{8 + i}"]) + rep(24, lambda i: [f"v_fma_f32 v{72 + i}, v8, v9, v10"]))
body("permlane32_swap", rep(16, lambda i: [f"v_permlane32_swap_b32 v{64 + 2 * i}, v{65 + 2 * i}"]))
body("permlane16_swap", rep(16, lambda i: [f"v_permlane16_swap_b32 v{64 + 2 * i}, v{65 + 2 * i}"]))
body("fmac dpp row_shl (independent)", rep(16, lambda i: [f"v_fmac_f32_dpp v{64 + i}, v{8 + i % 8}, v{16 + i % 8} row_shl:1 row_mask:0xf bank_mask:0xf"]))
body("mov dpp row_shr", rep(16, lambda i: [f"v_mov_b32_dpp v{64 + i}, v{8 + i % 8} row_shr:1 row_mask:0xf bank_mask:0xf"]))
body("v_cvt / shifts (bf16 widen: lshlrev, and)", rep(16, lambda i: [f"v_lshlrev_b32 v{64 + 2 * i}, 16, v{8 + i % 8}", f"v_and_b32 v{65 + 2 * i}, s20, v{8 + i % 8}"]))


# ---- the backward scan's state body, as issued (profiles/r05_issue_budget.md): 8 elements of one lane, one state ----
def bwd_state(dep):
    """dep = True: the kernel's dependent chains (3 recurrences of 8, the suffix scan); False: same opcodes, independent."""
    o = []
    dl, dlu, dy = 8, 16, 24            # f2 x 4 each: v[8:15], v[16:23], v[24:31]
    Bn, Cn = 32, 40                    # fp32 B / C of the state (from LDS in the kernel)
    a, xs, c, ax = 48, 56, 64, 72      # a2, xs2, c2, ax2
    S1, S2 = 80, 88
    vb, vc = 96, 104
    An2, Ar2, dA2 = 112, 114, 116
    tmp = 120
    for k in range(4):
        o.append(f"v_pk_mul_f32 v[{a + 2 * k}:{a + 2 * k + 1}], v[{dl + 2 * k}:{dl + 2 * k + 1}], v[{An2}:{An2 + 1}]")
    for i in range(8): o.append(f"v_exp_f32 v{a + i}, v{a + i}")
    for k in range(4):
        o.append(f"v_pk_mul_f32 v[{xs + 2 * k}:{xs + 2 * k + 1}], v[{dlu + 2 * k}:{dlu + 2 * k + 1}], v[{Bn + 2 * k}:{Bn + 2 * k + 1}]")
        o.append(f"v_pk_mul_f32 v[{c + 2 * k}:{c + 2 * k + 1}], v[{Cn + 2 * k}:{Cn + 2 * k + 1}], v[{dy + 2 * k}:{dy + 2 * k + 1}]")
    # rg chain (8 fma), ra: exp + mul + fma
    o.append(f"v_mov_b32 v{tmp}, 0")
    for i in range(7, -1, -1):
        o.append(f"v_fma_f32 v{tmp}, v{a + i}, v{tmp}, v{c + i}" if dep else f"v_fma_f32 v{tmp + 1 + i % 4}, v{a + i}, v{tmp}, v{c + i}")
    o.append(f"v_mul_f32 v{tmp + 5}, v{dl}, v{An2}")
    o.append(f"v_exp_f32 v{tmp + 5}, v{tmp + 5}")
    o.append(f"v_mul_f32 v{tmp + 5}, v{tmp + 5}, v{a}")
    o.append(f"v_fmac_f32 v{tmp}, v{tmp + 5}, v{tmp + 6}")
    # suffix scan: 4 steps x (fmac dpp, mul dpp, s_nop 0) as row_scan_suffix_b
    o.append("s_nop 1")
    for s in (1, 2, 4, 8):
        o.append(f"v_fmac_f32_dpp v{tmp}, v{tmp}, v{tmp + 5} row_shl:{s} row_mask:0xf bank_mask:0xf")
        o.append(f"v_mul_f32_dpp v{tmp + 5}, v{tmp + 5}, v{tmp + 5} row_shl:{s} row_mask:0xf bank_mask:0xf")
        o.append("s_nop 0")
    o.append("s_nop 0")
    o.append(f"v_mov_b32_dpp v{tmp + 7}, v{tmp} row_shl:1 row_mask:0xf bank_mask:0xf")
    # x recurrence: ax = a * xrun ; xrun = ax + b
    for i in range(8):
        src = tmp + 6 if (i == 0 or not dep) else xs + i - 1
        o.append(f"v_mul_f32 v{ax + i}, v{a + i}, v{src}")
        o.append(f"v_add_f32 v{xs + i}, v{ax + i}, v{xs + i}")
    # g recurrence
    for i in range(7, -1, -1):
        src = tmp + 7 if (i == 7 or not dep) else c + i + 1
        o.append(f"v_fma_f32 v{c + i}, v{a + (i + 1) % 8}, v{src}, v{c + i}")
    for k in range(4):
        o.append(f"v_pk_mul_f32 v[{tmp + 8}:{tmp + 9}], v[{c + 2 * k}:{c + 2 * k + 1}], v[{ax + 2 * k}:{ax + 2 * k + 1}]")   # gax
        o.append(f"v_pk_fma_f32 v[{S1 + 2 * k}:{S1 + 2 * k + 1}], v[{c + 2 * k}:{c + 2 * k + 1}], v[{Bn + 2 * k}:{Bn + 2 * k + 1}], v[{S1 + 2 * k}:{S1 + 2 * k + 1}]")
        o.append(f"v_pk_fma_f32 v[{S2 + 2 * k}:{S2 + 2 * k + 1}], v[{Ar2}:{Ar2 + 1}], v[{tmp + 8}:{tmp + 9}], v[{S2 + 2 * k}:{S2 + 2 * k + 1}]")
        o.append(f"v_pk_fma_f32 v[{dA2}:{dA2 + 1}], v[{dl + 2 * k}:{dl + 2 * k + 1}], v[{tmp + 8}:{tmp + 9}], v[{dA2}:{dA2 + 1}]")
        o.append(f"v_pk_mul_f32 v[{vb + 2 * k}:{vb + 2 * k + 1}], v[{c + 2 * k}:{c + 2 * k + 1}], v[{dlu + 2 * k}:{dlu + 2 * k + 1}]")
        o.append(f"v_pk_mul_f32 v[{vc + 2 * k}:{vc + 2 * k + 1}], v[{dy + 2 * k}:{dy + 2 * k + 1}], v[{xs + 2 * k}:{xs + 2 * k + 1}]")
    # dA row all-sum: add + 4 dpp adds
    o.append(f"v_add_f32 v{tmp + 10}, v{dA2}, v{dA2 + 1}")
    for s in (1, 2, 4, 8):
        o.append("s_nop 1")
        o.append(f"v_add_f32_dpp v{tmp + 10}, v{tmp + 10}, v{tmp + 10} row_ror:{s} row_mask:0xf bank_mask:0xf")
    # 4-row reduce-scatter of dB / dC: 8 permlane32 swaps, 4 pk_add, 4 permlane16 swaps, 2 pk_add
    o.append("s_nop 1")
    for i in range(8): o.append(f"v_permlane32_swap_b32 v{vb + i}, v{vc + i}")
    for k in range(4): o.append(f"v_pk_add_f32 v[{vb + 2 * k}:{vb + 2 * k + 1}], v[{vb + 2 * k}:{vb + 2 * k + 1}], v[{vc + 2 * k}:{vc + 2 * k + 1}]")
    o.append("s_nop 1")
    for i in range(4): o.append(f"v_permlane16_swap_b32 v{vb + i}, v{vb + 4 + i}")
    for k in range(2): o.append(f"v_pk_add_f32 v[{vb + 2 * k}:{vb + 2 * k + 1}], v[{vb + 2 * k}:{vb + 2 * k + 1}], v[{vb + 4 + 2 * k}:{vb + 5 + 2 * k}]")
    return o


body("BWD state body, kernel's chains", bwd_state(True))
body("BWD state body, chains cut", bwd_state(False))


def clobbers(ins):
    regs = set()
    for x in ins:
        for m in re.finditer(r"\bv(\d+)\b", x): regs.add(int(m.group(1)))
        for m in re.finditer(r"\bv\[(\d+):(\d+)\]", x): regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return ", ".join(f'"v{r}"' for r in sorted(regs))


print(r'''// GENERATED by tools/microbench/gen_mix_occ.py -- do not edit
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__device__ unsigned long long g_clk[2];
extern __shared__ float dyn_lds[];
template <int V>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    if (iters < 0) dyn_lds[threadIdx.x] = 1.f;
    // finite, small operands everywhere (a scan's value range): v8..v127 = 0.5 + lane * 2^-12
    asm volatile("v_cvt_f32_u32 v8, v0\n v_mul_f32 v8, 0x39800000, v8\n v_add_f32 v8, 0.5, v8\n"''')
for r in range(9, 128): print(f'        "v_mov_b32 v{r}, v8\\n"')
print('        "s_mov_b32 s20, 0x3f000000\\n s_mov_b32 s21, 0x3f000000\\n" ::: ' + ", ".join(f'"v{r}"' for r in range(8, 128)) + ', "s20", "s21");')
print("    for (int it = 0; it < iters; ++it) {")
for v, (name, ins) in enumerate(bodies):
    print(f"        if (V == {v}) asm volatile(")
    for x in ins: print(f'            "{x}\\n"')
    print(f"            ::: {clobbers(ins)});")
print(r'''    }
    float r;
    asm volatile("v_mov_b32 %0, v64" : "=v"(r));
    if (r == 12345.678f) out[0] = r;
    if (threadIdx.x == 0 && blockIdx.x == 7) { g_clk[0] = __builtin_readcyclecounter() - t0; g_clk[1] = wall_clock64() - w0; }
}

template <int V>
int run(const char* name, int n_instr, int n_valu) {
    float* d;
    CHECK(hipMalloc(&d, 4));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    printf("%-44s (%3d VALU)", name, n_valu);
    for (int occ : {1, 2, 3, 4, 8}) {
        const int iters = 6000 / occ + 200;
        const size_t smem = occ == 8 ? 0 : (160 * 1024 / occ) - 1024;   // occ workgroups of 4 waves per CU = occ waves per SIMD
        dim3 grid(256 * occ), block(256);
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<V>, grid, block, smem, 0, d, iters);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<V>, grid, block, smem, 0, d, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long hc[2] = {0, 0};
        CHECK(hipMemcpyFromSymbol(hc, HIP_SYMBOL(g_clk), 16));
        const double clk = hc[1] ? 100.0e6 * hc[0] / hc[1] : 2.4e9;
        const double cyc = ms * 1e-3 * clk / ((double)iters * occ);   // cycles of one SIMD per body
        printf("  occ%d %6.2f", occ, cyc / n_valu);
    }
    printf("   cycles per VALU instruction\n");
    CHECK(hipFree(d));
    return 0;
}

int main() {''')
for v, (name, ins) in enumerate(bodies):
    nv = sum(1 for x in ins if x.startswith("v_"))
    print(f'    run<{v}>("{name}", {len(ins)}, {nv});')
print("    return 0;\n}")
