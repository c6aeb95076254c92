/*
 * vms_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's selective-scan and causal-conv1d
 * algorithms.  It is the checker for the HIP path; it is never shipped, never
 * imported by the product packages and never the thing measured (only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load it).
 *
 * Parity pinning: every function here is checked in tests/test_oracle_golden.py
 * against golden vectors generated in the build container by importing the
 * reference's own pure-PyTorch path (tests/golden/make_golden.py):
 *   selective_scan_ref  mamba/mamba_ssm/ops/selective_scan_interface.py:86-152
 *   causal_conv1d_ref   causal-conv1d/causal_conv1d/causal_conv1d_interface.py:49-65
 *   causal_conv1d_update_ref                                    ...:87-104
 * and (for gradients) against torch.autograd through those functions.
 *
 * The file is compiled twice (oracle/Makefile): REAL=float (same arithmetic
 * width as the reference, which upcasts everything with .float()) and
 * REAL=double (a tighter "truth" used when judging fp32 kernels).
 * All arrays are dense, row-major, already widened to float by the caller;
 * rounding of outputs to a 16-bit I/O type is done by the caller.
 *
 * Threading: OpenMP over the independent (batch, dim) rows when built with
 * -fopenmp (bench.py's cpu_baseline states the thread count it used).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL float
#endif
#ifndef SUFFIX
#define SUFFIX f32
#endif
#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUFFIX)

typedef REAL real;

static inline real softplus_ref(real x) {
    /* F.softplus(beta=1, threshold=20): selective_scan_interface.py:106-107;
     * same threshold as the kernel (selective_scan_fwd_kernel.cuh:153-156). */
    return x <= (real)20 ? (real)log1p(exp((double)x)) : x;
}
static inline real sigmoid_ref(real x) { return (real)(1.0 / (1.0 + exp(-(double)x))); }

/* ------------------------------------------------------------------------- *
 * selective scan forward.  selective_scan_ref, SSI:86-152:
 *   delta = softplus(delta + delta_bias)           (:104-107)
 *   x_l   = exp(delta_l A_n) x_{l-1} + delta_l B_{n,l} u_l   (:121-134)
 *   y_l   = sum_n C_{n,l} x_{l,n}                  (:135-141)
 *   out   = y + u D ; out_z = out * silu(z)        (:148-150)
 * Shapes: u, delta, z, out, out_z (batch, dim, L); A (dim, N);
 *   variable B/C: (batch, G, N, L); constant B/C: (dim, N); D, delta_bias (dim).
 * x_ckpt (optional): (batch, dim, n_chunks, 2N) with n_chunks = ceil(L/2048):
 *   slot [c][2n+1] = state after the last element of 2048-chunk c
 *   (selective_scan_fwd_kernel.cuh:251-254, last_state = x[:, :, -1, 1::2] SSI:40);
 *   slot [c][2n]   = state after the first 1024 elements of chunk c (the HIP
 *   path's mid-chunk checkpoint; see DESIGN.md "x layout").
 * last_state (optional): (batch, dim, N).
 * ------------------------------------------------------------------------- */
void FN(vms_oracle_scan_fwd)(int batch, int dim, int L, int N, int G,
                             const float *u, const float *delta, const float *A,
                             const float *Bm, const float *Cm, const float *Dv,
                             const float *z, const float *delta_bias,
                             int var_B, int var_C, int delta_softplus,
                             float *out, float *out_z, float *x_ckpt, float *last_state) {
    const int n_chunks = (L + 2047) / 2048;
    const int dpg = dim / G;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < batch; ++b) {
        for (int d = 0; d < dim; ++d) {
            const int g = d / dpg;
            const float *ur = u + ((size_t)b * dim + d) * L;
            const float *dr = delta + ((size_t)b * dim + d) * L;
            const float *zr = z ? z + ((size_t)b * dim + d) * L : NULL;
            float *outr = out + ((size_t)b * dim + d) * L;
            float *ozr = out_z ? out_z + ((size_t)b * dim + d) * L : NULL;
            real *st = (real *)calloc((size_t)N, sizeof(real));
            const real bias = delta_bias ? (real)delta_bias[d] : (real)0;
            const real Dd = Dv ? (real)Dv[d] : (real)0;
            for (int l = 0; l < L; ++l) {
                real dl = (real)dr[l] + bias;
                if (delta_softplus) dl = softplus_ref(dl);
                const real ul = (real)ur[l];
                real y = 0;
                for (int n = 0; n < N; ++n) {
                    const real An = (real)A[(size_t)d * N + n];
                    const real Bn = var_B ? (real)Bm[(((size_t)b * G + g) * N + n) * L + l]
                                          : (real)Bm[(size_t)d * N + n];
                    const real Cn = var_C ? (real)Cm[(((size_t)b * G + g) * N + n) * L + l]
                                          : (real)Cm[(size_t)d * N + n];
                    const real a = (real)exp((double)(dl * An));
                    st[n] = a * st[n] + dl * Bn * ul;
                    y += st[n] * Cn;
                }
                real o = y + ul * Dd;
                outr[l] = (float)o;
                if (zr) {
                    const real zl = (real)zr[l];
                    ozr[l] = (float)(o * zl * sigmoid_ref(zl));
                }
                if (x_ckpt) {
                    const int c = l / 2048, r = l % 2048;
                    float *xc = x_ckpt + (((size_t)b * dim + d) * n_chunks + c) * 2 * N;
                    if (r == 1023 || (l == L - 1 && r < 1023))
                        for (int n = 0; n < N; ++n) xc[2 * n] = (float)st[n];
                    if (r == 2047 || l == L - 1)
                        for (int n = 0; n < N; ++n) xc[2 * n + 1] = (float)st[n];
                }
            }
            if (last_state)
                for (int n = 0; n < N; ++n) last_state[((size_t)b * dim + d) * N + n] = (float)st[n];
            free(st);
        }
    }
}

/* ------------------------------------------------------------------------- *
 * selective scan backward: the analytic gradient of the function above with
 * respect to (u, delta, A, B, C, D, z, delta_bias) for an upstream gradient
 * dout of the FINAL output (out_z when z is given, else out).  Formulas follow
 * selective_scan_bwd_kernel.cuh:171-206 (z gate), :244-329 (state adjoint
 * g_l = C_l dy_l + a_{l+1} g_{l+1}; du, ddelta, dA, dB, dC), :439-452 (softplus
 * chain) and are pinned against torch.autograd through selective_scan_ref.
 * dB / dC for variable B/C are (batch, G, N, L) fp32 sums over the dims of a
 * group; for constant B/C they are (dim, N).  dA (dim, N), dD, ddelta_bias
 * (dim) are sums over batch.  All outputs must be zero-initialised by the
 * caller where they are accumulated (dA, dB, dC, dD, ddelta_bias).
 * ------------------------------------------------------------------------- */
void FN(vms_oracle_scan_bwd)(int batch, int dim, int L, int N, int G,
                             const float *u, const float *delta, const float *A,
                             const float *Bm, const float *Cm, const float *Dv,
                             const float *z, const float *delta_bias, const float *dout,
                             int var_B, int var_C, int delta_softplus,
                             float *du, float *ddelta, float *dA, float *dB, float *dC,
                             float *dD, float *dz, float *ddelta_bias) {
    const int dpg = dim / G;
    /* accumulate shared outputs in double/real scratch to stay order-independent */
    const size_t nBC = (size_t)batch * G * N * L;
    real *dBs = var_B ? (real *)calloc(nBC, sizeof(real)) : NULL;
    real *dCs = var_C ? (real *)calloc(nBC, sizeof(real)) : NULL;
    real *dAs = (real *)calloc((size_t)dim * N, sizeof(real));
    real *dBc = !var_B ? (real *)calloc((size_t)dim * N, sizeof(real)) : NULL;
    real *dCc = !var_C ? (real *)calloc((size_t)dim * N, sizeof(real)) : NULL;
    real *dDs = (real *)calloc((size_t)dim, sizeof(real));
    real *dbs = (real *)calloc((size_t)dim, sizeof(real));
#pragma omp parallel for schedule(static)
    for (int d = 0; d < dim; ++d) {
        /* dims are the parallel axis so that per-dim sums need no atomics; the
         * (batch,G,N,L) sums over dims of a group are done under a critical
         * section per row below. */
        const int g = d / dpg;
        real *xs = (real *)malloc((size_t)L * N * sizeof(real)); /* x_{l,n} */
        real *dl_ = (real *)malloc((size_t)L * sizeof(real));
        real *gst = (real *)malloc((size_t)N * sizeof(real));
        real *rowB = var_B ? (real *)malloc((size_t)N * L * sizeof(real)) : NULL;
        real *rowC = var_C ? (real *)malloc((size_t)N * L * sizeof(real)) : NULL;
        for (int b = 0; b < batch; ++b) {
            const float *ur = u + ((size_t)b * dim + d) * L;
            const float *dr = delta + ((size_t)b * dim + d) * L;
            const float *zr = z ? z + ((size_t)b * dim + d) * L : NULL;
            const float *gor = dout + ((size_t)b * dim + d) * L;
            const real bias = delta_bias ? (real)delta_bias[d] : (real)0;
            const real Dd = Dv ? (real)Dv[d] : (real)0;
            /* forward recompute, keeping all states */
            for (int n = 0; n < N; ++n) gst[n] = 0;
            for (int l = 0; l < L; ++l) {
                real dl = (real)dr[l] + bias;
                if (delta_softplus) dl = softplus_ref(dl);
                dl_[l] = dl;
                for (int n = 0; n < N; ++n) {
                    const real An = (real)A[(size_t)d * N + n];
                    const real Bn = var_B ? (real)Bm[(((size_t)b * G + g) * N + n) * L + l]
                                          : (real)Bm[(size_t)d * N + n];
                    gst[n] = (real)exp((double)(dl * An)) * gst[n] + dl * Bn * (real)ur[l];
                    xs[(size_t)l * N + n] = gst[n];
                }
            }
            for (int n = 0; n < N; ++n) gst[n] = 0; /* now the adjoint a_{l+1} g_{l+1} */
            for (int l = L - 1; l >= 0; --l) {
                const real ul = (real)ur[l], dl = dl_[l];
                real dy = (real)gor[l];
                if (zr) {
                    /* y (pre-gate output) is needed for dz */
                    real y = ul * Dd;
                    for (int n = 0; n < N; ++n) {
                        const real Cn = var_C ? (real)Cm[(((size_t)b * G + g) * N + n) * L + l]
                                              : (real)Cm[(size_t)d * N + n];
                        y += xs[(size_t)l * N + n] * Cn;
                    }
                    const real zl = (real)zr[l], s = sigmoid_ref(zl);
                    dz[((size_t)b * dim + d) * L + l] = (float)(dy * y * s * ((real)1 + zl * ((real)1 - s)));
                    dy = dy * zl * s;
                }
                real dul = Dd * dy, ddl = 0;
                dDs[d] += dy * ul;
                for (int n = 0; n < N; ++n) {
                    const real An = (real)A[(size_t)d * N + n];
                    const real Bn = var_B ? (real)Bm[(((size_t)b * G + g) * N + n) * L + l]
                                          : (real)Bm[(size_t)d * N + n];
                    const real Cn = var_C ? (real)Cm[(((size_t)b * G + g) * N + n) * L + l]
                                          : (real)Cm[(size_t)d * N + n];
                    const real x = xs[(size_t)l * N + n];
                    const real gx = Cn * dy + gst[n]; /* dL/dx_{l,n} */
                    const real a = (real)exp((double)(dl * An));
                    const real ax = x - dl * Bn * ul; /* a_l x_{l-1} */
                    dul += gx * dl * Bn;
                    ddl += gx * Bn * ul + gx * An * ax;
                    dAs[(size_t)d * N + n] += gx * dl * ax;
                    if (var_B) rowB[(size_t)n * L + l] = gx * dl * ul;
                    else dBc[(size_t)d * N + n] += gx * dl * ul;
                    if (var_C) rowC[(size_t)n * L + l] = dy * x;
                    else dCc[(size_t)d * N + n] += dy * x;
                    gst[n] = a * gx;
                }
                du[((size_t)b * dim + d) * L + l] = (float)dul;
                if (delta_softplus) {
                    const real raw = (real)dr[l] + bias;
                    if (raw <= (real)20) ddl = ddl * sigmoid_ref(raw);
                }
                ddelta[((size_t)b * dim + d) * L + l] = (float)ddl;
                dbs[d] += ddl;
            }
            if (var_B || var_C) {
#pragma omp critical
                {
                    for (int n = 0; n < N; ++n)
                        for (int l = 0; l < L; ++l) {
                            const size_t o = (((size_t)b * G + g) * N + n) * L + l;
                            if (var_B) dBs[o] += rowB[(size_t)n * L + l];
                            if (var_C) dCs[o] += rowC[(size_t)n * L + l];
                        }
                }
            }
        }
        free(xs); free(dl_); free(gst); free(rowB); free(rowC);
    }
    for (size_t i = 0; i < (size_t)dim * N; ++i) dA[i] = (float)dAs[i];
    if (var_B) for (size_t i = 0; i < nBC; ++i) dB[i] = (float)dBs[i];
    else for (size_t i = 0; i < (size_t)dim * N; ++i) dB[i] = (float)dBc[i];
    if (var_C) for (size_t i = 0; i < nBC; ++i) dC[i] = (float)dCs[i];
    else for (size_t i = 0; i < (size_t)dim * N; ++i) dC[i] = (float)dCc[i];
    if (dD) for (int d = 0; d < dim; ++d) dD[d] = (float)dDs[d];
    if (ddelta_bias) for (int d = 0; d < dim; ++d) ddelta_bias[d] = (float)dbs[d];
    free(dBs); free(dCs); free(dAs); free(dBc); free(dCc); free(dDs); free(dbs);
}

/* ------------------------------------------------------------------------- *
 * causal depthwise conv1d forward.  causal_conv1d_ref, CCI:49-65:
 *   out[b,d,l] = act(bias_d + sum_{w<W} weight[d,w] * x[b,d,l-(W-1-w)]),
 * zero left padding (F.conv1d(padding=W-1)[..., :L]), act = silu or identity.
 * ------------------------------------------------------------------------- */
void FN(vms_oracle_conv_fwd)(int batch, int dim, int L, int W, const float *x,
                             const float *weight, const float *bias, int silu, float *out) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < batch; ++b)
        for (int d = 0; d < dim; ++d) {
            const float *xr = x + ((size_t)b * dim + d) * L;
            float *o = out + ((size_t)b * dim + d) * L;
            for (int l = 0; l < L; ++l) {
                real acc = bias ? (real)bias[d] : (real)0;
                for (int w = 0; w < W; ++w) {
                    const int s = l - (W - 1 - w);
                    if (s >= 0) acc += (real)weight[(size_t)d * W + w] * (real)xr[s];
                }
                o[l] = (float)(silu ? acc * sigmoid_ref(acc) : acc);
            }
        }
}

/* causal conv1d backward: gradient of the above for upstream dout.  Formulas as
 * causal_conv1d_bwd.cu:153-164 (silu'), :197-204 (dx), :225-239 (dweight, dbias);
 * pinned against autograd through causal_conv1d_ref. */
void FN(vms_oracle_conv_bwd)(int batch, int dim, int L, int W, const float *x,
                             const float *weight, const float *bias, const float *dout, int silu,
                             float *dx, float *dweight, float *dbias) {
#pragma omp parallel for schedule(static)
    for (int d = 0; d < dim; ++d) {
        real dw[8] = {0}, db = 0;
        real *g = (real *)malloc((size_t)L * sizeof(real));
        for (int b = 0; b < batch; ++b) {
            const float *xr = x + ((size_t)b * dim + d) * L;
            const float *gor = dout + ((size_t)b * dim + d) * L;
            for (int l = 0; l < L; ++l) {
                real go = (real)gor[l];
                if (silu) {
                    real pre = bias ? (real)bias[d] : (real)0;
                    for (int w = 0; w < W; ++w) {
                        const int s = l - (W - 1 - w);
                        if (s >= 0) pre += (real)weight[(size_t)d * W + w] * (real)xr[s];
                    }
                    const real sg = sigmoid_ref(pre);
                    go = go * sg * ((real)1 + pre * ((real)1 - sg));
                }
                g[l] = go;
                db += go;
                for (int w = 0; w < W; ++w) {
                    const int s = l - (W - 1 - w);
                    if (s >= 0) dw[w] += (real)xr[s] * go;
                }
            }
            for (int l = 0; l < L; ++l) {
                real acc = 0;
                for (int w = 0; w < W; ++w) {
                    const int t = l + (W - 1 - w);
                    if (t < L) acc += (real)weight[(size_t)d * W + w] * g[t];
                }
                dx[((size_t)b * dim + d) * L + l] = (float)acc;
            }
        }
        for (int w = 0; w < W; ++w) dweight[(size_t)d * W + w] = (float)dw[w];
        if (dbias) dbias[d] = (float)db;
        free(g);
    }
}

/* single-token decode step.  causal_conv1d_update_ref, CCI:87-104:
 * roll conv_state left by one, append x, dot with weight, bias, act.
 * conv_state (batch, dim, W) is updated in place; x, out (batch, dim). */
void FN(vms_oracle_conv_update)(int batch, int dim, int W, const float *x, float *conv_state,
                                const float *weight, const float *bias, int silu, float *out) {
    for (int b = 0; b < batch; ++b)
        for (int d = 0; d < dim; ++d) {
            float *cs = conv_state + ((size_t)b * dim + d) * W;
            for (int w = 0; w + 1 < W; ++w) cs[w] = cs[w + 1];
            cs[W - 1] = x[(size_t)b * dim + d];
            real acc = bias ? (real)bias[d] : (real)0;
            for (int w = 0; w < W; ++w) acc += (real)cs[w] * (real)weight[(size_t)d * W + w];
            out[(size_t)b * dim + d] = (float)(silu ? acc * sigmoid_ref(acc) : acc);
        }
}

/* ---- fused residual-add + LayerNorm / RMSNorm -------------------------------------------------------
 * layer_norm_ref / rms_norm_ref (mamba/mamba_ssm/ops/triton/layernorm.py:19-48) with upcast=True, which is
 * what the reference's fused kernels compute (fp32 statistics on x + residual, :85-120):
 *   s = x + residual ; [mean = sum(s)/N] ; var = sum((s - mean)^2)/N  (rms: sum(s^2)/N) ;
 *   rstd = 1/sqrt(var + eps) ; y = (s - mean) * rstd * w + b   (rms: s * rstd * w + b)
 * x, residual, y, res_out: (rows, cols); weight, bias: (cols); mean, rstd: (rows).  residual, bias,
 * res_out, mean may be NULL. */
void FN(vms_oracle_norm_fwd)(int rows, int cols, const float *x, const float *residual, const float *weight,
                             const float *bias, float eps, int is_rms, float *y, float *res_out, float *mean,
                             float *rstd) {
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r) {
        const float *xr = x + (size_t)r * cols;
        const float *rr = residual ? residual + (size_t)r * cols : NULL;
        real m = 0, v = 0;
        for (int c = 0; c < cols; ++c) m += (real)xr[c] + (rr ? (real)rr[c] : (real)0);
        m = is_rms ? (real)0 : m / cols;
        for (int c = 0; c < cols; ++c) {
            const real s = (real)xr[c] + (rr ? (real)rr[c] : (real)0) - m;
            v += s * s;
        }
        const real rs = (real)(1.0 / sqrt((double)(v / cols) + (double)eps));
        for (int c = 0; c < cols; ++c) {
            const real s = (real)xr[c] + (rr ? (real)rr[c] : (real)0);
            if (res_out) res_out[(size_t)r * cols + c] = (float)s;
            y[(size_t)r * cols + c] = (float)((s - m) * rs * (real)weight[c] + (bias ? (real)bias[c] : (real)0));
        }
        if (mean) mean[r] = (float)m;
        rstd[r] = (float)rs;
    }
}

/* gradient of the above w.r.t. the pre-norm sum s (= dx = dresidual when both share a dtype), weight and
 * bias; formulas of _layer_norm_bwd_kernel (layernorm.py:240-288):
 *   xhat = (s - mean) * rstd ; wdy = w * dy ; c1 = sum(xhat * wdy)/N ; c2 = sum(wdy)/N  (rms: c2 = 0, mean = 0)
 *   ds = (wdy - (xhat * c1 + c2)) * rstd [+ dres_out] ; dw = sum_rows dy * xhat ; db = sum_rows dy
 * s: (rows, cols) the saved pre-norm sum; dres_out (gradient of the prenorm output) may be NULL. */
void FN(vms_oracle_norm_bwd)(int rows, int cols, const float *s, const float *weight, const float *mean,
                             const float *rstd, const float *dy, const float *dres_out, int is_rms, float *ds,
                             float *dw, float *db) {
    real *dwa = (real *)calloc(cols, sizeof(real));
    real *dba = (real *)calloc(cols, sizeof(real));
    for (int r = 0; r < rows; ++r) {
        const float *sr = s + (size_t)r * cols, *dyr = dy + (size_t)r * cols;
        const real m = (is_rms || !mean) ? (real)0 : (real)mean[r], rs = (real)rstd[r];
        real c1 = 0, c2 = 0;
        for (int c = 0; c < cols; ++c) {
            const real xhat = ((real)sr[c] - m) * rs, wdy = (real)weight[c] * (real)dyr[c];
            c1 += xhat * wdy;
            c2 += wdy;
            dwa[c] += (real)dyr[c] * xhat;
            dba[c] += (real)dyr[c];
        }
        c1 /= cols;
        c2 = is_rms ? (real)0 : c2 / cols;
        for (int c = 0; c < cols; ++c) {
            const real xhat = ((real)sr[c] - m) * rs, wdy = (real)weight[c] * (real)dyr[c];
            real g = (wdy - (xhat * c1 + c2)) * rs;
            if (dres_out) g += (real)dres_out[(size_t)r * cols + c];
            ds[(size_t)r * cols + c] = (float)g;
        }
    }
    for (int c = 0; c < cols; ++c) {
        dw[c] = (float)dwa[c];
        if (db) db[c] = (float)dba[c];
    }
    free(dwa);
    free(dba);
}

/* ---- single-token SSM step -------------------------------------------------------------------------
 * selective_state_update_ref (mamba/mamba_ssm/ops/triton/selective_state_update.py:157-192):
 *   dt = softplus?(dt + dt_bias) ; state = state * exp(dt * A) + dt * B * x ; out = sum_n state * C + D * x ;
 *   out *= silu(z).   state (batch, dim, N) is updated in place; x, dt, z, out (batch, dim); A (dim, N);
 *   B, C (batch, N); D, dt_bias (dim).  D, z, dt_bias may be NULL. */
void FN(vms_oracle_state_update)(int batch, int dim, int N, float *state, const float *x, const float *dt,
                                 const float *A, const float *B, const float *C, const float *D, const float *z,
                                 const float *dt_bias, int dt_softplus, float *out) {
    for (int b = 0; b < batch; ++b)
        for (int d = 0; d < dim; ++d) {
            real t = (real)dt[(size_t)b * dim + d] + (dt_bias ? (real)dt_bias[d] : (real)0);
            if (dt_softplus) t = softplus_ref(t);
            const real xv = (real)x[(size_t)b * dim + d];
            float *st = state + ((size_t)b * dim + d) * N;
            real acc = 0;
            for (int n = 0; n < N; ++n) {
                const real s = (real)st[n] * (real)exp((double)(t * (real)A[(size_t)d * N + n])) +
                               t * (real)B[(size_t)b * N + n] * xv;
                st[n] = (float)s;
                acc += (real)st[n] * (real)C[(size_t)b * N + n];
            }
            if (D) acc += xv * (real)D[d];
            if (z) {
                const real zv = (real)z[(size_t)b * dim + d];
                acc *= zv * sigmoid_ref(zv);
            }
            out[(size_t)b * dim + d] = (float)acc;
        }
}

/* ------------------------------------------------------------------------- *
 * selective scan with COMPLEX A (the reference's weight_t = complex<float> instantiations,
 * selective_scan.cpp:282-287; semantics = selective_scan_ref's complex branch, SSI:111-116, 144-145):
 *   a_l = exp(delta_l A_n) (complex), x_l = a_l x_{l-1} + delta_l u_l B_{n,l}, y_l = 2 Re(sum_n C_{n,l} x_{l,n}).
 * Complex numbers are (re, im) float pairs: A, constant B / C: (dim, N, 2); variable B / C: (batch, G, N, L, 2)
 * = the reference's real (batch, G, N, 2L) tensors; x_ckpt: (batch, dim, n_chunks, 2N, 2) with the slots of the
 * real function; last_state (batch, dim, N, 2).
 * ------------------------------------------------------------------------- */
typedef struct { real re, im; } cplx;
static inline cplx c_mul(cplx a, cplx b) { cplx r = {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; return r; }
static inline cplx c_conj(cplx a) { cplx r = {a.re, -a.im}; return r; }
static inline cplx c_exp_scaled(real dl, cplx A) {
    const real m = (real)exp((double)(dl * A.re));
    cplx r = {m * (real)cos((double)(dl * A.im)), m * (real)sin((double)(dl * A.im))};
    return r;
}
static inline cplx c_ld(const float *p, size_t i) { cplx r = {(real)p[2 * i], (real)p[2 * i + 1]}; return r; }

void FN(vms_oracle_cscan_fwd)(int batch, int dim, int L, int N, int G,
                              const float *u, const float *delta, const float *A,
                              const float *Bm, const float *Cm, const float *Dv,
                              const float *z, const float *delta_bias,
                              int var_B, int var_C, int delta_softplus,
                              float *out, float *out_z, float *x_ckpt, float *last_state) {
    const int n_chunks = (L + 2047) / 2048;
    const int dpg = dim / G;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < batch; ++b) {
        for (int d = 0; d < dim; ++d) {
            const int g = d / dpg;
            const size_t row = ((size_t)b * dim + d) * L;
            cplx *st = (cplx *)calloc((size_t)N, sizeof(cplx));
            const real bias = delta_bias ? (real)delta_bias[d] : (real)0;
            const real Dd = Dv ? (real)Dv[d] : (real)0;
            for (int l = 0; l < L; ++l) {
                real dl = (real)delta[row + l] + bias;
                if (delta_softplus) dl = softplus_ref(dl);
                const real ul = (real)u[row + l];
                real y = 0;
                for (int n = 0; n < N; ++n) {
                    const cplx An = c_ld(A, (size_t)d * N + n);
                    const cplx Bn = var_B ? c_ld(Bm, (((size_t)b * G + g) * N + n) * L + l) : c_ld(Bm, (size_t)d * N + n);
                    const cplx Cn = var_C ? c_ld(Cm, (((size_t)b * G + g) * N + n) * L + l) : c_ld(Cm, (size_t)d * N + n);
                    const cplx a = c_exp_scaled(dl, An);
                    cplx s = c_mul(a, st[n]);
                    s.re += dl * ul * Bn.re;
                    s.im += dl * ul * Bn.im;
                    st[n] = s;
                    y += (real)2 * (s.re * Cn.re - s.im * Cn.im);
                }
                const real o = y + ul * Dd;
                out[row + l] = (float)o;
                if (z) {
                    const real zl = (real)z[row + l];
                    out_z[row + l] = (float)(o * zl * sigmoid_ref(zl));
                }
                if (x_ckpt) {
                    const int c = l / 2048, r = l % 2048;
                    float *xc = x_ckpt + (((size_t)b * dim + d) * n_chunks + c) * 4 * N;
                    if (r == 1023 || (l == L - 1 && r < 1023))
                        for (int n = 0; n < N; ++n) { xc[4 * n] = (float)st[n].re; xc[4 * n + 1] = (float)st[n].im; }
                    if (r == 2047 || l == L - 1)
                        for (int n = 0; n < N; ++n) { xc[4 * n + 2] = (float)st[n].re; xc[4 * n + 3] = (float)st[n].im; }
                }
            }
            if (last_state)
                for (int n = 0; n < N; ++n) {
                    last_state[(((size_t)b * dim + d) * N + n) * 2] = (float)st[n].re;
                    last_state[(((size_t)b * dim + d) * N + n) * 2 + 1] = (float)st[n].im;
                }
            free(st);
        }
    }
}

/* Backward of the function above; gradients of complex parameters in PyTorch's convention (dL/dRe + i dL/dIm), which is
 * what the reference's kernels return (selective_scan_bwd_kernel.cuh:330-436) and what torch.autograd gives through
 * selective_scan_ref (the fixtures):
 *   g_l = 2 dy_l conj(C_l) + conj(a_{l+1}) g_{l+1};   dC_l = 2 dy_l conj(x_l);   dB_l = delta_l u_l g_l;
 *   du_l = D dy_l + delta_l sum_n Re(conj(B) g);   ddelta_l = sum_n u_l Re(conj(B) g) + Re(conj(A a_l x_{l-1}) g);
 *   dA_n = sum_l delta_l conj(a_l x_{l-1}) g_l.
 * dA, constant dB / dC: (dim, N, 2); variable dB / dC: (batch, G, N, L, 2). */
void FN(vms_oracle_cscan_bwd)(int batch, int dim, int L, int N, int G,
                              const float *u, const float *delta, const float *A,
                              const float *Bm, const float *Cm, const float *Dv,
                              const float *z, const float *delta_bias, const float *dout,
                              int var_B, int var_C, int delta_softplus,
                              float *du, float *ddelta, float *dA, float *dB, float *dC,
                              float *dD, float *dz, float *ddelta_bias) {
    const int dpg = dim / G;
    const size_t nBC = (size_t)batch * G * N * L;
    cplx *dBs = var_B ? (cplx *)calloc(nBC, sizeof(cplx)) : NULL;
    cplx *dCs = var_C ? (cplx *)calloc(nBC, sizeof(cplx)) : NULL;
    cplx *dAs = (cplx *)calloc((size_t)dim * N, sizeof(cplx));
    cplx *dBc = !var_B ? (cplx *)calloc((size_t)dim * N, sizeof(cplx)) : NULL;
    cplx *dCc = !var_C ? (cplx *)calloc((size_t)dim * N, sizeof(cplx)) : NULL;
    real *dDs = (real *)calloc((size_t)dim, sizeof(real));
    real *dbs = (real *)calloc((size_t)dim, sizeof(real));
#pragma omp parallel for schedule(static)
    for (int d = 0; d < dim; ++d) {
        const int g = d / dpg;
        cplx *xs = (cplx *)malloc((size_t)L * N * sizeof(cplx));
        real *dl_ = (real *)malloc((size_t)L * sizeof(real));
        cplx *gst = (cplx *)malloc((size_t)N * sizeof(cplx));
        cplx *rowB = var_B ? (cplx *)malloc((size_t)N * L * sizeof(cplx)) : NULL;
        cplx *rowC = var_C ? (cplx *)malloc((size_t)N * L * sizeof(cplx)) : NULL;
        for (int b = 0; b < batch; ++b) {
            const size_t row = ((size_t)b * dim + d) * L;
            const real bias = delta_bias ? (real)delta_bias[d] : (real)0;
            const real Dd = Dv ? (real)Dv[d] : (real)0;
            for (int n = 0; n < N; ++n) { gst[n].re = 0; gst[n].im = 0; }
            for (int l = 0; l < L; ++l) {
                real dl = (real)delta[row + l] + bias;
                if (delta_softplus) dl = softplus_ref(dl);
                dl_[l] = dl;
                for (int n = 0; n < N; ++n) {
                    const cplx An = c_ld(A, (size_t)d * N + n);
                    const cplx Bn = var_B ? c_ld(Bm, (((size_t)b * G + g) * N + n) * L + l) : c_ld(Bm, (size_t)d * N + n);
                    cplx s = c_mul(c_exp_scaled(dl, An), gst[n]);
                    s.re += dl * (real)u[row + l] * Bn.re;
                    s.im += dl * (real)u[row + l] * Bn.im;
                    gst[n] = s;
                    xs[(size_t)l * N + n] = s;
                }
            }
            for (int n = 0; n < N; ++n) { gst[n].re = 0; gst[n].im = 0; } /* conj(a_{l+1}) g_{l+1} */
            for (int l = L - 1; l >= 0; --l) {
                const real ul = (real)u[row + l], dl = dl_[l];
                real dy = (real)dout[row + l];
                if (z) {
                    real y = ul * Dd;
                    for (int n = 0; n < N; ++n) {
                        const cplx Cn = var_C ? c_ld(Cm, (((size_t)b * G + g) * N + n) * L + l) : c_ld(Cm, (size_t)d * N + n);
                        const cplx x = xs[(size_t)l * N + n];
                        y += (real)2 * (x.re * Cn.re - x.im * Cn.im);
                    }
                    const real zl = (real)z[row + l], s = sigmoid_ref(zl);
                    dz[row + l] = (float)(dy * y * s * ((real)1 + zl * ((real)1 - s)));
                    dy = dy * zl * s;
                }
                real dul = Dd * dy, ddl = 0;
                dDs[d] += dy * ul;
                for (int n = 0; n < N; ++n) {
                    const cplx An = c_ld(A, (size_t)d * N + n);
                    const cplx Bn = var_B ? c_ld(Bm, (((size_t)b * G + g) * N + n) * L + l) : c_ld(Bm, (size_t)d * N + n);
                    const cplx Cn = var_C ? c_ld(Cm, (((size_t)b * G + g) * N + n) * L + l) : c_ld(Cm, (size_t)d * N + n);
                    const cplx x = xs[(size_t)l * N + n];
                    cplx gx = {(real)2 * dy * Cn.re + gst[n].re, -(real)2 * dy * Cn.im + gst[n].im};
                    const cplx a = c_exp_scaled(dl, An);
                    const cplx ax = {x.re - dl * ul * Bn.re, x.im - dl * ul * Bn.im}; /* a_l x_{l-1} */
                    const real bg = Bn.re * gx.re + Bn.im * gx.im;                   /* Re(conj(B) g) */
                    const cplx Aax = c_mul(An, ax);
                    dul += dl * bg;
                    ddl += ul * bg + (Aax.re * gx.re + Aax.im * gx.im);
                    const cplx cag = c_mul(c_conj(ax), gx);
                    dAs[(size_t)d * N + n].re += dl * cag.re;
                    dAs[(size_t)d * N + n].im += dl * cag.im;
                    const cplx dBv = {dl * ul * gx.re, dl * ul * gx.im};
                    const cplx dCv = {(real)2 * dy * x.re, -(real)2 * dy * x.im};
                    if (var_B) rowB[(size_t)n * L + l] = dBv;
                    else { dBc[(size_t)d * N + n].re += dBv.re; dBc[(size_t)d * N + n].im += dBv.im; }
                    if (var_C) rowC[(size_t)n * L + l] = dCv;
                    else { dCc[(size_t)d * N + n].re += dCv.re; dCc[(size_t)d * N + n].im += dCv.im; }
                    gst[n] = c_mul(c_conj(a), gx);
                }
                du[row + l] = (float)dul;
                if (delta_softplus) {
                    const real raw = (real)delta[row + l] + bias;
                    if (raw <= (real)20) ddl = ddl * sigmoid_ref(raw);
                }
                ddelta[row + l] = (float)ddl;
                dbs[d] += ddl;
            }
            if (var_B || var_C) {
#pragma omp critical
                {
                    for (int n = 0; n < N; ++n)
                        for (int l = 0; l < L; ++l) {
                            const size_t o = (((size_t)b * G + g) * N + n) * L + l;
                            if (var_B) { dBs[o].re += rowB[(size_t)n * L + l].re; dBs[o].im += rowB[(size_t)n * L + l].im; }
                            if (var_C) { dCs[o].re += rowC[(size_t)n * L + l].re; dCs[o].im += rowC[(size_t)n * L + l].im; }
                        }
                }
            }
        }
        free(xs); free(dl_); free(gst); free(rowB); free(rowC);
    }
    for (size_t i = 0; i < (size_t)dim * N; ++i) { dA[2 * i] = (float)dAs[i].re; dA[2 * i + 1] = (float)dAs[i].im; }
    const size_t nB = var_B ? nBC : (size_t)dim * N, nC = var_C ? nBC : (size_t)dim * N;
    const cplx *srcB = var_B ? dBs : dBc, *srcC = var_C ? dCs : dCc;
    for (size_t i = 0; i < nB; ++i) { dB[2 * i] = (float)srcB[i].re; dB[2 * i + 1] = (float)srcB[i].im; }
    for (size_t i = 0; i < nC; ++i) { dC[2 * i] = (float)srcC[i].re; dC[2 * i + 1] = (float)srcC[i].im; }
    if (dD) for (int d = 0; d < dim; ++d) dD[d] = (float)dDs[d];
    if (ddelta_bias) for (int d = 0; d < dim; ++d) ddelta_bias[d] = (float)dbs[d];
    free(dBs); free(dCs); free(dAs); free(dBc); free(dCc); free(dDs); free(dbs);
}
