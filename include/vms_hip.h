/*
 * vms_hip.h -- C ABI of libvms_hip.so, the MI355X (gfx950) selective-scan /
 * causal-conv1d hot path of the Video Mamba Suite.
 *
 * This is the drop-in boundary.  Each entry point replaces one function of the
 * reference's two pybind extensions (paths relative to the reference tree):
 *
 *   vms_selective_scan_fwd   <- selective_scan_cuda.fwd
 *        mamba/csrc/selective_scan/selective_scan.cpp:226-336  (host)
 *        mamba/csrc/selective_scan/selective_scan_fwd_kernel.cuh:67-345
 *   vms_selective_scan_bwd   <- selective_scan_cuda.bwd
 *        mamba/csrc/selective_scan/selective_scan.cpp:338-492
 *        mamba/csrc/selective_scan/selective_scan_bwd_kernel.cuh:75-531
 *   vms_causal_conv1d_fwd    <- causal_conv1d_cuda.causal_conv1d_fwd
 *        causal-conv1d/csrc/causal_conv1d.cpp:130-189, causal_conv1d_fwd.cu
 *   vms_causal_conv1d_bwd    <- causal_conv1d_cuda.causal_conv1d_bwd
 *        causal-conv1d/csrc/causal_conv1d.cpp:191-268, causal_conv1d_bwd.cu
 *   vms_causal_conv1d_update <- causal_conv1d_cuda.causal_conv1d_update
 *        causal-conv1d/csrc/causal_conv1d.cpp:270-327, causal_conv1d_update.cu
 *   vms_selective_state_update <- the Triton kernel of mamba_ssm.ops.triton.selective_state_update
 *        mamba/mamba_ssm/ops/triton/selective_state_update.py:16-154
 *   vms_layer_norm_fwd / _bwd <- the Triton kernels of mamba_ssm.ops.triton.layernorm
 *        mamba/mamba_ssm/ops/triton/layernorm.py:51-377
 *   vms_proj_apply / vms_proj_wgrad / vms_proj_kred <- the small x_proj / dt_proj GEMMs MambaInnerFn runs through torch
 *        mamba/mamba_ssm/ops/selective_scan_interface.py:182, 275-279
 *
 * Conventions
 *   - plain C: POD parameter blocks, raw device pointers, sizes; no torch types.
 *   - every stride is in ELEMENTS of the tensor's own dtype and 64-bit (the
 *     reference stores 32-bit strides, selective_scan.h:27).
 *   - the library never allocates, never synchronises, reads no environment
 *     variable and keeps no mutable state: all outputs (and accumulators that must
 *     start at zero, marked [zeroed]) are provided by the caller, exactly the
 *     tensors the reference host functions allocate; every choice a call makes
 *     follows from its parameter block.  What it caches is immutable per-device
 *     information, filled once per device under std::call_once (the CU count,
 *     and the one-time hipFuncSetAttribute that admits > 64 KB of dynamic LDS).
 *     Calls are re-entrant from any number of host threads, on any device (the
 *     current device of the calling thread, as for the reference's extensions,
 *     selective_scan.cpp:326-327) and are enqueued on `stream` (a hipStream_t
 *     passed as void*; NULL = the null stream).
 *   - return value: VMS_OK or a negative vms_status; vms_last_error() gives a
 *     thread-local message naming the failed check (the reference raises
 *     RuntimeError from TORCH_CHECK with the failing expression).
 */
#ifndef VMS_HIP_H
#define VMS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VMS_ABI_VERSION 11

typedef enum {
    VMS_OK = 0,
    VMS_ERR_INVALID_ARG = -1,   /* a shape / dtype / stride / NULL check failed          */
    VMS_ERR_UNSUPPORTED = -2,   /* valid in the reference but not built here (complex A)  */
    VMS_ERR_LAUNCH = -3         /* hipGetLastError() after the launch was not hipSuccess  */
} vms_status;

typedef enum { VMS_F32 = 0, VMS_F16 = 1, VMS_BF16 = 2 } vms_dtype;

/* kernel generations of the selective scan: a call runs the highest one that is <= `impl` and eligible for the
 * problem; AUTO = PAIR.  GENERIC takes every problem the reference takes; PAIR is the fast paths (variable B / C,
 * dstate 4 / 8 / 16, ...).  ABI v11: the values 2 (FAST), 4 (ROWS) -- experimental generations of rounds 1-3, aliases since
 * round 5 -- and 5 (OCC4: round 5's 128-VGPR two-direction backward, measured 12-20 % slower than the default and kept as
 * profiles/r05_bwd_occ4.patch + profiles/r05_bwd_occ4.md) are gone: every kernel in the library is reachable by the default
 * dispatch or is a generic one.  They are rejected as out of range. */
typedef enum { VMS_IMPL_AUTO = 0, VMS_IMPL_GENERIC = 1, VMS_IMPL_PAIR = 3 } vms_scan_impl;
#define VMS_BUILD_EXPERIMENTAL 1   /* bit of vms_build_flags(): never set since round 5 (the FAST / ROWS / MFMA generations are gone) */

/* ---- selective scan ------------------------------------------------------------------
 * u, delta, z, out, out_z : (batch, dim, seqlen), unit seqlen stride, free batch/dim strides
 * A                       : (dim, dstate) fp32; with is_complex (ABI v8) complex64, see the field below
 * B, C variable           : (batch, n_groups, dstate, seqlen) in `dtype`, unit seqlen stride
 * B, C constant           : (dim, dstate) fp32
 * D, delta_bias           : (dim) fp32 or NULL
 * x                       : (batch, dim, n_chunks, 2*dstate) fp32 contiguous,
 *                           n_chunks = ceil(seqlen / 2048) (selective_scan.cpp:307,313).
 *                           slot [c][2n+1] = state after chunk c (as the reference;
 *                           last_state = x[:, :, -1, 1::2]); slot [c][2n] = state after the
 *                           first 1024 elements of chunk c (the reference stores the running
 *                           product of exp(delta*A) there, which nothing reads back).
 */
typedef struct {
    int32_t batch, dim, seqlen, dstate, n_groups, n_chunks;
    int32_t dtype;            /* vms_dtype of u, delta, z, out, out_z and variable B/C */
    int32_t is_variable_B, is_variable_C, delta_softplus;
    const void *u, *delta, *A, *B, *C, *D, *z, *delta_bias;
    void *out, *out_z, *x;
    int64_t u_batch_stride, u_d_stride;
    int64_t delta_batch_stride, delta_d_stride;
    int64_t z_batch_stride, z_d_stride;
    int64_t out_batch_stride, out_d_stride;
    int64_t out_z_batch_stride, out_z_d_stride;
    int64_t A_d_stride, A_dstate_stride;
    int64_t B_batch_stride, B_group_stride, B_d_stride, B_dstate_stride;
    int64_t C_batch_stride, C_group_stride, C_d_stride, C_dstate_stride;
    /* x row pitch: elements between x[b,d,c,:] and x[b,d,c+1,:]; 0 means 2*dstate (dense).
     * x_has_sub != 0: the pitch is >= 18*dstate and x[b,d,c, 2*dstate + s*dstate + n], s = 0..15,
     * holds the state after the first 128*(s+1) elements of chunk c (finer checkpoints that let the
     * backward kernel walk 128-element chunks; they live in the same allocation, behind the
     * reference-shaped (.., 2*dstate) view the Python layer hands out). */
    int64_t x_chunk_stride;
    /* x_has_sub == 2 (the row-major layout of the experimental kernels of rounds 1-3): rejected since round 5. */
    /* x_has_sub == 3 ("lane checkpoints", ABI v7): the pitch is >= 258*dstate (even; x 8-byte aligned) and
     * x[b,d,c, 2*dstate + ((n/4)*256 + i)*4 + n%4], i = 0..255, holds state n after the first 8*(i+1) elements of chunk c,
     * in scan order (16x the data of x_has_sub == 1, of which every 16th entry is the same 128-element checkpoint).
     * The forward writes them from values it already has; the whole-vector backward kernel (dstate 16, variable B / C,
     * seqlen % 8 == 0) then takes the state entering each lane's 8 elements from x instead of rebuilding it (-10 % of its
     * instructions: 897 -> 805 us at (8, 1024, 8192); the forward pays 35-40 us for the 537 MB of stores).
     * vms_scan_x_pitch() says when the library wants it. */
    int32_t x_has_sub;
    /* reverse != 0 (an extension; the reference has no such flag): the scan runs right-to-left,
     * i.e. the call equals flip(op(flip(every seqlen-indexed tensor))) without the copies the
     * reference's bidirectional blocks pay (mamba_simple.py:244,258, mamba_new.py:193,213).
     * x keeps scan order (chunk c = the c-th 2048 elements visited). */
    int32_t reverse;
    /* out_z_accumulate != 0 (an extension): out_z += instead of out_z = ; the caller's out_z already holds the
     * other direction's gated output of a bidirectional block (z must be given; the "rows" kernels decline). */
    int32_t out_z_accumulate;
    /* bc_pad (an extension): the caller guarantees that every B and C row (variable B / C) is readable and FINITE for
     * bc_pad elements past its logical end -- after element seqlen-1, or before element 0 when reverse != 0 (e.g. zero
     * padding to the next multiple of 16).  With it the fast kernels also take seqlen % 16 != 0 (they read B / C in
     * 16-byte vectors); 0 = no guarantee, such lengths run on the generic kernels. */
    int32_t bc_pad;
    /* optional scratch for the fast kernels (vms_scan_fwd_workspace_bytes / _bwd_; 16-byte aligned,
     * contents undefined on entry and exit, private to this call until the stream reaches its end).
     * NULL / too small -> the kernels that need none run.  Forward: only the opt-in "rows" kernels use it.
     * Backward: with few rows and long sequences (e.g. batch 1, 768 channels, 65,536 tokens = 24 workgroups for 256 CUs)
     * the paired kernel splits every row into up to 16 ranges of chunks; the scratch holds the (P, q) adjoint carries
     * that chain them (vms_scan_bwd_workspace_bytes() > 0 exactly when it wants to split). */
    void *workspace;
    int64_t workspace_bytes;
    /* ABI v4: the knobs that used to be environment variables read inside the library.  impl: one of vms_scan_impl, 0 = the
     * default.  segments: 0 = the library decides whether to split long rows of a small grid into ranges of chunks
     * (from the device's CU count); n >= 1 forces n ranges (1 = never split) -- tests and profiling. */
    int32_t impl;
    int32_t segments;
    /* ABI v5 (an extension): reverse_from > 0 makes the direction a property of the BATCH ENTRY -- entries
     * [0, reverse_from) run left-to-right, entries [reverse_from, batch) right-to-left; `reverse` must then be 0.  One call
     * serves both directions of a block whose directions share their weights (the DBM block, mamba_new.py:192-213, which
     * stacks the flipped half on the batch axis: here the second half of the batch is simply scanned the other way, no
     * copies).  Results per entry are exactly those of two calls on the two sub-batches.  0 = `reverse` decides. */
    int32_t reverse_from;
    /* ABI v8: is_complex != 0 = the reference's weight_t = complex<float> instantiations (selective_scan.cpp:47, 282-287;
     * semantics of selective_scan_ref's complex branch, SSI:111-116, 144-145: y = 2 Re(sum_n C x) + D u).  A, constant B / C
     * (and dA, constant dB / dC) are complex64 = (re, im) float pairs, their strides counted in COMPLEX elements; variable
     * B / C are the reference's real (batch, n_groups, dstate, 2*seqlen) tensors of interleaved pairs in `dtype` (strides
     * in elements of `dtype`, dB / dC fp32 of the same shape); x is complex64 (batch, dim, n_chunks, 2*dstate) with the slots
     * of the real case, x_chunk_stride in complex elements (0 = 2*dstate).  x_has_sub == 1: the pitch is >= 6*dstate and
     * x[b,d,c, 2*dstate + s*dstate + n], s = 0..3, holds state n after the first 512*(s+1) elements of chunk c -- what the
     * backward starts its 512-element chunks from (required by vms_selective_scan_bwd when seqlen > 512); other x_has_sub
     * values, reverse_from and bc_pad do not apply.  Gradients of complex parameters follow PyTorch's convention
     * (dL/dRe + i dL/dIm), as the reference's kernels return them.  Served by its own kernels (selective_scan_complex.hip,
     * built like the generic real ones: no suite model has a complex A). */
    int32_t is_complex;
} vms_scan_fwd_params;

/* backward.  dout is the gradient of the final output (out_z when z != NULL, else out).
 * out (the pre-gate forward output) is required iff z != NULL (selective_scan.cpp:424-429).
 * x is required iff n_chunks > 1 (:449).  dz: required iff z != NULL (may be a view into a
 * larger tensor, SSI:244-248).  out_z: optional recompute target (:440-442).
 * dA (dim,dstate) fp32 [zeroed]; dB, dC: variable -> (batch,n_groups,dstate,seqlen) FP32
 * [zeroed] (the reference accumulates in fp32 and casts afterwards, :461-462, 488);
 * constant -> (dim,dstate) fp32 [zeroed]; dD, ddelta_bias (dim) fp32 [zeroed] or NULL. */
typedef struct {
    vms_scan_fwd_params f;    /* f.out = pre-gate out (read), f.out_z = optional recompute (write) */
    const void *dout;
    void *du, *ddelta, *dz;
    float *dA, *dB, *dC, *dD, *ddelta_bias;
    int64_t dout_batch_stride, dout_d_stride;
    int64_t du_batch_stride, du_d_stride;
    int64_t ddelta_batch_stride, ddelta_d_stride;
    int64_t dz_batch_stride, dz_d_stride;
    int64_t dA_d_stride, dA_dstate_stride;
    int64_t dB_batch_stride, dB_group_stride, dB_d_stride, dB_dstate_stride;
    int64_t dC_batch_stride, dC_group_stride, dC_d_stride, dC_dstate_stride;
    /* dz_accumulate != 0 (an extension): dz += instead of dz = ; the caller's dz already holds the gradient
     * that reached z by another path (the other direction of a bidirectional block).  32-bit pad follows. */
    int32_t dz_accumulate;
    int32_t reserved0;
} vms_scan_bwd_params;

int vms_selective_scan_fwd(const vms_scan_fwd_params *p, void *stream);
int vms_selective_scan_bwd(const vms_scan_bwd_params *p, void *stream);
/* ABI v9 (an extension): the backward scans of BOTH directions of a bidirectional block (mamba_simple.py:234-258: two
 * parameter sets over the same rows, the reference launches selective_scan_bwd twice, SSI:541-561) in one call.
 * Result = vms_selective_scan_bwd(a) followed by vms_selective_scan_bwd(b) with b's dz ADDED to a's: every gradient of a and
 * of b in its own tensors, the gradient z receives through both directions in a->dz (b->dz: NULL or the same tensor;
 * a->dz_accumulate as usual).  When the pair qualifies -- a left-to-right, b right-to-left, the same z and dout, equal
 * sizes / dtype / checkpoint layout, whole-vector rows, together at least one workgroup per CU, no forced split, a->dz_accumulate
 * == 0 -- it is
 * ONE grid: a's workgroups write dz = dout (out_a + out_b) dsilu(z) (linear in the pre-gate outputs; rounded once, where
 * the two-call form rounds the first direction's part to `dtype` before adding the second's), b's none; otherwise the two calls run
 * back to back.  A direction of the suite's most common shape, (8, 768, 3136), is 192 workgroups for 256 CUs; both
 * directions as one grid of 4-wave workgroups take 489 us instead of 2 x 317.  vms_scan_bwd_dual_fused() != 0: the pair
 * qualifies (then the workspaces of a and b are not used). */
int vms_selective_scan_bwd_dual(const vms_scan_bwd_params *a, const vms_scan_bwd_params *b, void *stream);
int vms_scan_bwd_dual_fused(const vms_scan_bwd_params *a, const vms_scan_bwd_params *b);
/* scratch the fast kernels want for this problem (0: none / not eligible); only sizes, dtype and
 * flags of *p are read */
int64_t vms_scan_fwd_workspace_bytes(const vms_scan_fwd_params *p);
int64_t vms_scan_bwd_workspace_bytes(const vms_scan_bwd_params *p);
/* batch*dim*n_chunks*2*dstate: the floats of the reference-shaped x (kept from the ABI of the row-major layout) */
int64_t vms_scan_x_elems(const vms_scan_fwd_params *p);
/* the x pitch (floats between x[b,d,c,:] and x[b,d,c+1,:]) to allocate for this problem; only sizes and flags of *p are
 * read.  mode 0: 2*dstate (the reference's tensor, selective_scan.cpp:313: no fast backward); 1: 18*dstate (x_has_sub == 1);
 * 3 or -1 (the library's choice for a forward whose backward will run): 258*dstate (x_has_sub == 3) when the backward
 * kernel that uses it takes the problem (variable B / C, dstate 16 -- or, ABI v11, 8 or 4 --, (dim / n_groups) % 32 == 0), seqlen % 16 == 0 (the
 * forward kernel that writes the checkpoints as whole lines) and x stays under 2 GiB per batch entry and 2^31 elements in all, else 18*dstate.  The caller sets x_chunk_stride to the
 * pitch and x_has_sub to 3 / 1 / 0 for pitch >= 258*dstate / >= 18*dstate / less.
 * Short sequences with many rows (seqlen <= 16, batch*dim >= 4096, variable B / C, dstate 16, (dim / n_groups) % 64 == 0: the suite's
 * TimeMamba scans along time) are served by lane-per-row kernels that keep no checkpoints (csrc/selective_scan_short.hip): every mode
 * answers 2*dstate for them (ABI v11, modes other than 0: 2*dstate + (ceil(seqlen / 16) - 1)*dstate for rows of 17 .. 64 elements, which run as chained 16-element
 * segments: x[b,d,0, 2*dstate + s*dstate + n] = state n after segment s), and with x_has_sub == 0 both vms_selective_scan_fwd and _bwd take that path (the backward rebuilds a row's
 * states and wants vms_scan_bwd_workspace_bytes() = batch*18*dim floats for its sums over the batch; without them it uses atomics). */
int64_t vms_scan_x_pitch(const vms_scan_fwd_params *p, int32_t mode);

/* ---- causal depthwise conv1d ---------------------------------------------------------
 * x, out, dout, dx : (batch, dim, seqlen); either unit seqlen stride (any batch/channel
 *                    stride) or unit channel stride ("channel-last", causal_conv1d.cpp:151-156)
 * weight           : (dim, width), 2 <= width <= 4 (:157); bias (dim) or NULL; both wdtype
 * dweight (dim,width), dbias (dim): FP32 [zeroed] accumulators (:247-249)
 * conv_state       : (batch, dim, width) in `dtype`, updated in place by _update
 */
typedef struct {
    int32_t batch, dim, seqlen, width;
    int32_t dtype;            /* vms_dtype of x, out, dout, dx, conv_state */
    int32_t wdtype;           /* vms_dtype of weight and bias              */
    int32_t silu_activation;
    int32_t reverse;          /* != 0: anti-causal (flip o conv o flip); seqlen-contiguous layout only */
    const void *x, *weight, *bias;
    void *out;
    int64_t x_batch_stride, x_c_stride, x_l_stride;
    int64_t weight_c_stride, weight_width_stride;
    int64_t out_batch_stride, out_c_stride, out_l_stride;
    /* update only */
    void *conv_state;
    int64_t conv_state_batch_stride, conv_state_c_stride, conv_state_l_stride;
    /* ABI v5: as vms_scan_fwd_params.reverse_from -- batch entries >= reverse_from are filtered anti-causally
     * (seqlen-contiguous layout only; `reverse` must be 0 when this is set) */
    int32_t reverse_from;
    int32_t reserved1;
} vms_conv_fwd_params;

typedef struct {
    vms_conv_fwd_params f;    /* f.out unused */
    const void *dout;
    void *dx;
    float *dweight, *dbias;
    int64_t dout_batch_stride, dout_c_stride, dout_l_stride;
    int64_t dx_batch_stride, dx_c_stride, dx_l_stride;
    int64_t dweight_c_stride, dweight_width_stride;
    /* dx_accumulate != 0 (an extension): dx += instead of dx = (see dz_accumulate) */
    int32_t dx_accumulate;
    int32_t reserved0;
} vms_conv_bwd_params;

int vms_causal_conv1d_fwd(const vms_conv_fwd_params *p, void *stream);
/* ABI v8 (an extension): both directions of a bidirectional block in ONE pass over x -- out = the causal filter (f.weight, f.bias),
 * out_b = the anti-causal filter (weight_b, bias_b) of the same rows, i.e. what vms_causal_conv1d_fwd gives for f and for
 * {f with reverse = 1, weight_b, bias_b, out_b} (same tap order: bit for bit for bf16 / fp32 I/O); the reference computes conv(x) and flip(conv_b(flip(x)))
 * (mamba_simple.py:244-258).  Seqlen-contiguous layout; f.reverse, f.reverse_from, f.conv_state must be 0; weight_b / bias_b
 * in f.wdtype with f.width taps. */
typedef struct {
    vms_conv_fwd_params f;
    const void *weight_b, *bias_b;
    void *out_b;
    int64_t weight_b_c_stride, weight_b_width_stride;
    int64_t out_b_batch_stride, out_b_c_stride;
} vms_conv_fwd_dual_params;
int vms_causal_conv1d_fwd_dual(const vms_conv_fwd_dual_params *p, void *stream);
int vms_sizeof_conv_fwd_dual_params(void);

int vms_causal_conv1d_bwd(const vms_conv_bwd_params *p, void *stream);
int vms_causal_conv1d_update(const vms_conv_fwd_params *p, void *stream);

/* ---- fused residual-add + LayerNorm / RMSNorm -------------------------------------------
 * replaces the Triton kernels behind layer_norm_fn / rms_norm_fn / RMSNorm
 * (mamba/mamba_ssm/ops/triton/layernorm.py: _layer_norm_fwd :122-173, _layer_norm_bwd :291-377).
 * x, y, residual, residual_out : (rows, cols), unit column stride; weight, bias : (cols) FP32
 * s = x + residual (fp32) ; y = (s - mean) * rstd * weight + bias   (is_rms: mean = 0)
 * residual_out (optional) receives s in res_dtype; mean (optional, unused for rms), rstd : (rows) fp32 */
typedef struct {
    int32_t rows, cols;
    int32_t x_dtype;          /* vms_dtype of x, y (forward) / dy, dx (backward)                       */
    int32_t res_dtype;        /* vms_dtype of residual, residual_out / s, dres_out, dres_in: x_dtype or VMS_F32 */
    int32_t is_rms;
    float eps;
    const void *x, *residual, *weight, *bias;
    void *y, *residual_out;
    float *mean, *rstd;
    int64_t x_row_stride, residual_row_stride, y_row_stride, residual_out_row_stride;
} vms_norm_params;

/* backward.  s = the forward's pre-norm sum (residual_out, or x when none was stored), dy = grad of y,
 * dres_out = grad of the prenorm output (optional).  dx (x_dtype); dres_in (res_dtype, optional: wanted
 * when the residual's dtype differs from x's, layernorm.py:331, 373-375).  dw_partial / db_partial:
 * (n_partials, cols) fp32, every row written; the caller sums over rows (the reference sums one partial
 * per SM, :339-343, 371-372).  n_partials = vms_layer_norm_bwd_partials(&f). */
typedef struct {
    vms_norm_params f;        /* rows, cols, dtypes, is_rms, weight, mean, rstd are read */
    const void *s, *dy, *dres_out;
    void *dx, *dres_in;
    float *dw_partial, *db_partial;
    int32_t n_partials, reserved;
    int64_t s_row_stride, dy_row_stride, dres_out_row_stride, dx_row_stride, dres_in_row_stride;
} vms_norm_bwd_params;

int vms_layer_norm_fwd(const vms_norm_params *p, void *stream);
int vms_layer_norm_bwd(const vms_norm_bwd_params *p, void *stream);
int vms_layer_norm_bwd_partials(const vms_norm_params *p);
/* ABI v10: dw[c] = sum over the n_partials rows of dw_partial (and db from db_partial, both or neither), rounded to out_dtype
 * (vms_dtype: the weight's) -- the sum the caller of vms_layer_norm_bwd owes, as one launch for both arrays.  cols % 4 == 0. */
int vms_layer_norm_bwd_finish(const float *dw_partial, const float *db_partial, int n_partials, int cols, void *dw, void *db,
                              int out_dtype, void *stream);

/* ---- single-token SSM step ------------------------------------------------------------------
 * replaces the Triton kernel behind selective_state_update
 * (mamba/mamba_ssm/ops/triton/selective_state_update.py:16-154).
 * state (batch, dim, dstate) in/out; x, out (batch, dim) in x_dtype; dt, z (batch, dim) in dt_dtype / z_dtype;
 * A (dim, dstate), D, dt_bias (dim) in w_dtype; B, C (batch, dstate) in bc_dtype; D, z, dt_bias may be NULL. */
typedef struct {
    int32_t batch, dim, dstate;
    int32_t state_dtype, x_dtype, bc_dtype, w_dtype;
    int32_t dt_softplus;
    int32_t dt_dtype, z_dtype;   /* ABI v4: dt and z in their own dtypes, as the reference's kernel loads them (:77-83) */
    void *state;
    const void *x, *dt, *A, *B, *C, *D, *z, *dt_bias;
    void *out;
    int64_t state_batch_stride, state_d_stride, state_n_stride;
    int64_t x_batch_stride, x_d_stride, dt_batch_stride, dt_d_stride, z_batch_stride, z_d_stride;
    int64_t out_batch_stride, out_d_stride;
    int64_t A_d_stride, A_n_stride, B_batch_stride, B_n_stride, C_batch_stride, C_n_stride;
} vms_state_update_params;

int vms_selective_state_update(const vms_state_update_params *p, void *stream);

/* ---- small projections of the inner node (ABI v6) ------------------------------------------------------
 * The skinny GEMMs between conv1d and the scan in MambaInnerFn(NoOutProj)
 * (mamba/mamba_ssm/ops/selective_scan_interface.py): forward
 *     delta = delta_proj_weight @ x_dbl[:R]                                  (:182)
 * backward
 *     ddelta_proj_weight = einsum("dB,Br->dr", ddelta, x_dbl[:, :R])         (:275)
 *     dx_dbl[:, :R]      = einsum("dB,dr->Br", ddelta, delta_proj_weight)    (:276)
 *     dx_proj_weight     = einsum("Br,Bd->rd", dx_dbl, conv1d_out)           (:278)
 *     dconv1d_out        = addmm(dconv1d_out, x_proj_weight.t(), dx_dbl.t()) (:279)
 * on this build's (batch, rows, seqlen) activations with a unit seqlen stride.  16-bit activations (bf16 / fp16), fp32
 * accumulation on the matrix cores; other dtypes / unaligned problems are left to the caller's library GEMM.
 *
 * vms_proj_apply:  out[b][d][l] (+)= sum_{r < k} w[d][r] * in[b][r][l]      d < rows, k <= 96
 *   w: (rows, k) in `dtype`, any strides;  in: (batch, k, seqlen);  out: (batch, rows, seqlen);
 *   accumulate != 0: out holds a value of `dtype` that the product is added to in fp32 before rounding (:279). */
typedef struct {
    int32_t batch, rows, k, seqlen;
    int32_t dtype;            /* vms_dtype of w, in, out: VMS_BF16 or VMS_F16 */
    int32_t accumulate;
    int32_t tiles_per_wg;     /* 64-position tiles a workgroup walks; 0 = chosen from the CU count (a tuning knob, like `segments`) */
    int32_t reserved;
    const void *w, *in;
    void *out;
    int64_t w_row_stride, w_k_stride;
    int64_t in_batch_stride, in_k_stride;          /* elements; multiples of 8 */
    int64_t out_batch_stride, out_row_stride;
} vms_proj_apply_params;

/* vms_proj_wgrad:  dw[m][n] += sum_{b, l} p[b][m][l] * q[b][n][l]           m <= 128
 *   p: (batch, m, seqlen), q: (batch, n, seqlen) in `dtype`; dw: (m, n) fp32, ADDED to with fp32 atomics (the caller
 *   zero-fills it, like the scan's dA / dD), so several calls may accumulate into one buffer. */
typedef struct {
    int32_t batch, m, n, seqlen;
    int32_t dtype;
    int32_t tiles_per_wg;     /* 64-position tiles per workgroup (each workgroup ends with m x 128 atomics); 0 = automatic */
    int32_t dw_transposed;    /* ABI v10: != 0: dw is stored (n, m) -- dw[n * dw_row_stride + m] -- e.g. ddt_proj.weight in the parameter's own layout */
    int32_t reserved;
    const void *p, *q;
    float *dw;
    int64_t p_batch_stride, p_row_stride, q_batch_stride, q_row_stride;   /* elements; multiples of 8 */
    int64_t dw_row_stride;
} vms_proj_wgrad_params;

int vms_proj_apply(const vms_proj_apply_params *p, void *stream);
int vms_proj_wgrad(const vms_proj_wgrad_params *p, void *stream);

/* vms_proj_kred (ABI v10): the inner node's two skinny products whose contraction runs over the CHANNELS --
 *     x_dbl        = F.linear(rearrange(conv1d_out, "b d l -> (b l) d"), x_proj_weight)   (:181)  m = dt_rank + 2 d_state
 *     dx_dbl[:, :R] = einsum("dB,dr->Br", ddelta, delta_proj_weight)                       (:276)  m = dt_rank
 *   out[b][m][l] = sum_{k < K} w[m][k] * in[b][k][l]          m <= 96, any K
 *   in: (batch, k, seqlen), out: (batch, m, seqlen) in `dtype` (bf16 / fp16), unit seqlen stride; w: (m, k) in `dtype` with a unit
 *   stride along k (x_proj.weight) OR along m (dt_proj.weight (d_inner, dt_rank) read as its transpose: w_row_stride = 1,
 *   w_k_stride = dt_rank).  fp32 accumulation on the matrix cores, one rounding to `dtype`.
 *   w2 / in2 / out2 (all or none): a second problem of the same shape and strides in the same launch -- the other direction of
 *   a bidirectional block. */
typedef struct {
    int32_t batch, m, k, seqlen;
    int32_t dtype;
    int32_t tile;             /* positions per workgroup: 64, 128 or 256; 0 = chosen from the grid size (a tuning knob) */
    const void *w, *in;
    void *out;
    const void *w2, *in2;
    void *out2;
    int64_t w_row_stride, w_k_stride;
    int64_t in_batch_stride, in_k_stride;          /* elements; multiples of 8 */
    int64_t out_batch_stride, out_row_stride;
    /* optional side job of the backward call (selective_scan_interface.py:258-271: dx_dbl[:, R:] = dB, dC): the fp32 sums the scan's
     * backward left in its [zeroed] accumulators are rounded to `dtype` into the rows of `out` that FOLLOW the m product rows --
     * out[b][m + g * cast_rows + n][l] = cast_src[g * cast_group_stride + b * cast_batch_stride + n * cast_row_stride + l],
     * g < cast_groups, n < cast_rows (unit seqlen stride, 16-byte aligned rows).  NULL = no such rows.  cast_src2: the same for out2. */
    const float *cast_src, *cast_src2;
    int32_t cast_rows, cast_groups;
    int64_t cast_group_stride, cast_batch_stride, cast_row_stride;
} vms_proj_kred_params;
int vms_proj_kred(const vms_proj_kred_params *p, void *stream);

/* vms_conv_xproj_dual (ABI v10): the head of a bidirectional block's forward in one pass over x --
 *     conv1d_out   = causal_conv1d_fwd(x, conv1d.weight, conv1d.bias, silu)            (:170-176)   for both parameter sets,
 *     x_dbl        = F.linear(rearrange(conv1d_out, "b d l -> (b l) d"), x_proj.weight)  (:181)       the second set right-to-left
 * c = the parameter block of vms_causal_conv1d_fwd_dual (out / out_b: the two conv1d outputs, bit-identical to that call's);
 * x_dbl / x_dbl_b (batch, m, seqlen) = w_x @ out / w_x_b @ out_b as vms_proj_kred computes them (w_x, w_x_b: (m, dim), unit stride
 * along dim, m <= 96).  16-bit activations (bf16 / fp16), conv weights in fp32 or the activations' dtype, SiLU on, unit seqlen
 * strides, seqlen / dim / strides multiples of 8; anything else: the two separate calls. */
typedef struct {
    vms_conv_fwd_dual_params c;
    const void *w_x, *w_x_b;
    void *x_dbl, *x_dbl_b;
    int32_t m;
    int32_t tile;             /* positions per workgroup: 64 or 128; 0 = chosen from the grid size */
    int64_t wx_row_stride;
    int64_t xdbl_batch_stride, xdbl_row_stride;
} vms_conv_xproj_dual_params;
int vms_conv_xproj_dual(const vms_conv_xproj_dual_params *p, void *stream);

/* vms_proj_conv_bwd: selective_scan_interface.py:278-283 in one pass over the activations --
 *     dx_proj_weight  = einsum("Br,Bd->rd", dx_dbl, conv1d_out)                      -> dw_x (k, dim) fp32, ADDED to (atomics)
 *     dconv1d_out     = addmm(du, x_proj_weight.t(), dx_dbl.t())                     (never stored: fp32 on chip)
 *     dx, dconv1d_weight, dconv1d_bias = causal_conv1d_bwd(x, w, b, dconv1d_out, dx, silu = True)
 * with conv1d_out recomputed from x exactly as vms_causal_conv1d_fwd computes it.  x (the conv's input), du (the scan's
 * gradient w.r.t. its input u), dx : (batch, dim, seqlen); dx_dbl : (batch, k, seqlen); all `dtype` (bf16 / fp16), unit
 * seqlen stride; w_x = x_proj.weight (k, dim) in `dtype`, any strides; conv weight (dim, width), bias (dim) or NULL in
 * `wdtype`; dconv_weight (dim, width), dconv_bias (dim), dw_x (k, dim): fp32 [zeroed] accumulators.  1 <= k <= 96 (k <= 32 since ABI v11).
 * seqlen % 8 == 0 needs 16-byte aligned bases and strides that are multiples of 8 elements (whole-vector pieces); any other
 * seqlen runs the ragged flavour (2-byte aligned rows, the row's last piece element by element).
 * reverse / reverse_from / dx_accumulate as in vms_conv_bwd_params. */
typedef struct {
    int32_t batch, dim, k, seqlen, width;
    int32_t dtype, wdtype;
    int32_t reverse, reverse_from, dx_accumulate;
    int32_t tiles_per_wg;     /* 0 = automatic */
    int32_t reserved;
    const void *x, *du, *dx_dbl, *w_x, *conv_weight, *conv_bias;
    void *dx;
    float *dconv_weight, *dconv_bias, *dw_x;
    int64_t x_batch_stride, x_c_stride;
    int64_t du_batch_stride, du_c_stride;
    int64_t dxdbl_batch_stride, dxdbl_k_stride;
    int64_t wx_k_stride, wx_c_stride;
    int64_t conv_weight_c_stride, conv_weight_width_stride;
    int64_t dx_batch_stride, dx_c_stride;
    int64_t dconv_weight_c_stride, dconv_weight_width_stride;
    int64_t dwx_k_stride;
} vms_proj_conv_bwd_params;
int vms_proj_conv_bwd(const vms_proj_conv_bwd_params *p, void *stream);

/* ---- misc ---------------------------------------------------------------------------- */
/* ---- per-step parameter preparation (ABI v8) -------------------------------------------------------------
 * What a block does to its PARAMETERS at the top of every training step, as one launch instead of one small kernel per
 * tensor: autocast's casts of the projection weights (selective_scan_interface.py:169-171 and F.linear's own), this build's
 * K-contiguous copy of in_proj's weight (a cast + transpose), A = -exp(A_log.float()) (mamba_simple.py:230, 246).
 * A job reads a (rows, cols) matrix with unit column stride and writes
 *   VMS_PREP_CAST     dst (rows, cols) = src converted to dst_dtype
 *   VMS_PREP_CAST_T   dst (cols, rows) = src^T converted to dst_dtype
 *   VMS_PREP_NEG_EXP  dst (rows, cols) = -exp(src) (fp32 arithmetic)
 * Row strides in elements; src and dst of different jobs may not overlap a job's dst (interleaved halves are disjoint elements). */
enum { VMS_PREP_CAST = 0, VMS_PREP_CAST_T = 1, VMS_PREP_NEG_EXP = 2 };
#define VMS_PREP_MAX_JOBS 8
typedef struct {
    const void *src;
    void *dst;
    int32_t rows, cols;
    int64_t src_row_stride, dst_row_stride;
    int32_t src_dtype, dst_dtype;   /* vms_dtype */
    int32_t op;
    int32_t dst_col_stride;         /* ABI v10: elements between consecutive destination columns; 0 = 1.  2 = one half of an
                                     * interleaved matrix (the DBM block's stacked projections, projections.py InProjFn / OutProjFn) */
} vms_prep_job;
typedef struct {
    int32_t n_jobs, reserved;
    vms_prep_job job[VMS_PREP_MAX_JOBS];
} vms_prep_params;
int vms_param_prep(const vms_prep_params *p, void *stream);
/* ABI v10: dst[i] = sum_{s < n_slices} src[s * slice_stride + i], i < n -- the sum over the K slices of a weight-gradient GEMM that
 * ran as a batched GEMM (this build's formulation of mamba_simple.py's in_proj / out_proj weight gradients: one workgroup per CU
 * instead of 32 output tiles); 16-bit slices (src_dtype), fp32 accumulation, dst in dst_dtype (the parameter's).  n, slice_stride
 * multiples of 8 elements. */
int vms_sum_slices(const void *src, int src_dtype, int n_slices, int64_t n, int64_t slice_stride, void *dst, int dst_dtype, void *stream);
int vms_sizeof_prep_params(void);

int vms_abi_version(void);
const char *vms_last_error(void);       /* thread-local, valid until the next failing call */
/* thread-local: the kernel family the last successful launch call of this thread enqueued, e.g. "scan_fwd_pair",
 * "scan_bwd_pair+split", "scan_fwd_generic", "conv_fwd_strips4", "conv_bwd_strips1", "conv_fwd_channel_last" --
 * lets a caller (and the tests) see when a problem was declined by a fast path */
const char *vms_last_kernel(void);
int vms_build_flags(void);              /* VMS_BUILD_* bits */
/* sizes of the parameter blocks as compiled, so a binding can verify its mirror */
int vms_sizeof_scan_fwd_params(void);
int vms_sizeof_scan_bwd_params(void);
int vms_sizeof_conv_fwd_params(void);
int vms_sizeof_conv_bwd_params(void);
int vms_sizeof_norm_params(void);
int vms_sizeof_norm_bwd_params(void);
int vms_sizeof_state_update_params(void);
int vms_sizeof_proj_apply_params(void);
int vms_sizeof_proj_wgrad_params(void);
int vms_sizeof_proj_conv_bwd_params(void);
int vms_sizeof_proj_kred_params(void);
int vms_sizeof_conv_xproj_dual_params(void);

#ifdef __cplusplus
}
#endif
#endif /* VMS_HIP_H */
