/* ORACLE -- test infrastructure only.  Never linked, imported or executed by the product path
 * (unet-studio_amd/, include/): only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may use it, and only as the checker.
 *
 * Plain-C restatement of the arithmetic the reference's hot path asks libtorch for.  The reference
 * (unet.cpp) holds no arithmetic of its own: each function below restates the published semantics
 * of the torch::nn module that the cited reference line instantiates (third-party dependency
 * libtorch; reference pins 2.0.0+cu117 / 1.13.0, image has 2.10.0), and is pinned in
 * tests/test_oracle.py against the ATen CPU kernels themselves and against tests/golden/.
 *
 * Layout: NCDHW with N == 1 (every forward of the reference is batch 1: train.cpp:615-621,
 * evaluate.cpp:226), fp32 storage, double accumulation.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define IDX3(z, y, x, H, W) (((int64_t)(z) * (H) + (y)) * (W) + (x))

/* ---- Conv3d, kernel ks in {1,3}, stride in {1,2}, padding (ks-1)/2, bias.  unet.cpp:59-72 ---- */
void orc_conv3d_fwd(const float* x, const float* w, const float* b, float* y, int Cin, int Cout, int D, int H, int W,
                    int ks, int stride) {
    int pad = (ks - 1) / 2;
    int Do = (D + 2 * pad - ks) / stride + 1, Ho = (H + 2 * pad - ks) / stride + 1, Wo = (W + 2 * pad - ks) / stride + 1;
    int64_t S = (int64_t)D * H * W, So = (int64_t)Do * Ho * Wo;
    int k3 = ks * ks * ks;
#pragma omp parallel for collapse(2) schedule(static)
    for (int co = 0; co < Cout; ++co)
        for (int z = 0; z < Do; ++z)
            for (int yy = 0; yy < Ho; ++yy)
                for (int xx = 0; xx < Wo; ++xx) {
                    double acc = b ? (double)b[co] : 0.0;
                    for (int ci = 0; ci < Cin; ++ci) {
                        const float* wp = w + ((int64_t)co * Cin + ci) * k3;
                        const float* xp = x + (int64_t)ci * S;
                        for (int kz = 0; kz < ks; ++kz) {
                            int iz = z * stride + kz - pad;
                            if (iz < 0 || iz >= D) continue;
                            for (int ky = 0; ky < ks; ++ky) {
                                int iy = yy * stride + ky - pad;
                                if (iy < 0 || iy >= H) continue;
                                for (int kx = 0; kx < ks; ++kx) {
                                    int ix = xx * stride + kx - pad;
                                    if (ix < 0 || ix >= W) continue;
                                    acc += (double)xp[IDX3(iz, iy, ix, H, W)] * (double)wp[(kz * ks + ky) * ks + kx];
                                }
                            }
                        }
                    }
                    y[(int64_t)co * So + IDX3(z, yy, xx, Ho, Wo)] = (float)acc;
                }
}

/* dL/dx of the conv above (autograd backward triggered at train.cpp:706). */
void orc_conv3d_bwd_data(const float* dy, const float* w, float* dx, int Cin, int Cout, int D, int H, int W, int ks,
                         int stride) {
    int pad = (ks - 1) / 2;
    int Do = (D + 2 * pad - ks) / stride + 1, Ho = (H + 2 * pad - ks) / stride + 1, Wo = (W + 2 * pad - ks) / stride + 1;
    int64_t S = (int64_t)D * H * W, So = (int64_t)Do * Ho * Wo;
    int k3 = ks * ks * ks;
#pragma omp parallel for collapse(2) schedule(static)
    for (int ci = 0; ci < Cin; ++ci)
        for (int iz = 0; iz < D; ++iz)
            for (int iy = 0; iy < H; ++iy)
                for (int ix = 0; ix < W; ++ix) {
                    double acc = 0.0;
                    for (int kz = 0; kz < ks; ++kz) {
                        int tz = iz + pad - kz;
                        if (tz < 0 || tz % stride) continue;
                        int z = tz / stride;
                        if (z >= Do) continue;
                        for (int ky = 0; ky < ks; ++ky) {
                            int ty = iy + pad - ky;
                            if (ty < 0 || ty % stride) continue;
                            int yy = ty / stride;
                            if (yy >= Ho) continue;
                            for (int kx = 0; kx < ks; ++kx) {
                                int tx = ix + pad - kx;
                                if (tx < 0 || tx % stride) continue;
                                int xx = tx / stride;
                                if (xx >= Wo) continue;
                                int64_t o = IDX3(z, yy, xx, Ho, Wo);
                                int t = (kz * ks + ky) * ks + kx;
                                for (int co = 0; co < Cout; ++co)
                                    acc += (double)dy[(int64_t)co * So + o] * (double)w[((int64_t)co * Cin + ci) * k3 + t];
                            }
                        }
                    }
                    dx[(int64_t)ci * S + IDX3(iz, iy, ix, H, W)] = (float)acc;
                }
}

/* dL/dw (accumulated: += as .grad does) and dL/db. */
void orc_conv3d_bwd_weight(const float* x, const float* dy, float* dw, float* db, int Cin, int Cout, int D, int H, int W,
                           int ks, int stride) {
    int pad = (ks - 1) / 2;
    int Do = (D + 2 * pad - ks) / stride + 1, Ho = (H + 2 * pad - ks) / stride + 1, Wo = (W + 2 * pad - ks) / stride + 1;
    int64_t S = (int64_t)D * H * W, So = (int64_t)Do * Ho * Wo;
    int k3 = ks * ks * ks;
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci)
            for (int kz = 0; kz < ks; ++kz)
                for (int ky = 0; ky < ks; ++ky)
                    for (int kx = 0; kx < ks; ++kx) {
                        double acc = 0.0;
                        for (int z = 0; z < Do; ++z) {
                            int iz = z * stride + kz - pad;
                            if (iz < 0 || iz >= D) continue;
                            for (int yy = 0; yy < Ho; ++yy) {
                                int iy = yy * stride + ky - pad;
                                if (iy < 0 || iy >= H) continue;
                                for (int xx = 0; xx < Wo; ++xx) {
                                    int ix = xx * stride + kx - pad;
                                    if (ix < 0 || ix >= W) continue;
                                    acc += (double)x[(int64_t)ci * S + IDX3(iz, iy, ix, H, W)] *
                                           (double)dy[(int64_t)co * So + IDX3(z, yy, xx, Ho, Wo)];
                                }
                            }
                        }
                        dw[((int64_t)co * Cin + ci) * k3 + (kz * ks + ky) * ks + kx] += (float)acc;
                    }
    if (db)
#pragma omp parallel for
        for (int co = 0; co < Cout; ++co) {
            double acc = 0.0;
            for (int64_t i = 0; i < So; ++i) acc += dy[(int64_t)co * So + i];
            db[co] += (float)acc;
        }
}

/* ---- ConvTranspose3d ks2 stride2 (the only legal form, unet.cpp:46-57); weight [Cin,Cout,2,2,2] ---- */
void orc_convt_fwd(const float* x, const float* w, const float* b, float* y, int Cin, int Cout, int D, int H, int W) {
    int Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
    int64_t S = (int64_t)D * H * W, So = (int64_t)Do * Ho * Wo;
#pragma omp parallel for collapse(2) schedule(static)
    for (int co = 0; co < Cout; ++co)
        for (int z = 0; z < Do; ++z)
            for (int yy = 0; yy < Ho; ++yy)
                for (int xx = 0; xx < Wo; ++xx) {
                    int t = ((z & 1) * 2 + (yy & 1)) * 2 + (xx & 1);
                    int64_t i = IDX3(z >> 1, yy >> 1, xx >> 1, H, W);
                    double acc = b ? (double)b[co] : 0.0;
                    for (int ci = 0; ci < Cin; ++ci)
                        acc += (double)x[(int64_t)ci * S + i] * (double)w[((int64_t)ci * Cout + co) * 8 + t];
                    y[(int64_t)co * So + IDX3(z, yy, xx, Ho, Wo)] = (float)acc;
                }
}

void orc_convt_bwd_data(const float* dy, const float* w, float* dx, int Cin, int Cout, int D, int H, int W) {
    int Ho = 2 * H, Wo = 2 * W;
    int64_t S = (int64_t)D * H * W, So = 8 * S;
#pragma omp parallel for collapse(2) schedule(static)
    for (int ci = 0; ci < Cin; ++ci)
        for (int z = 0; z < D; ++z)
            for (int yy = 0; yy < H; ++yy)
                for (int xx = 0; xx < W; ++xx) {
                    double acc = 0.0;
                    for (int t = 0; t < 8; ++t) {
                        int64_t o = IDX3(2 * z + (t >> 2), 2 * yy + ((t >> 1) & 1), 2 * xx + (t & 1), Ho, Wo);
                        for (int co = 0; co < Cout; ++co)
                            acc += (double)dy[(int64_t)co * So + o] * (double)w[((int64_t)ci * Cout + co) * 8 + t];
                    }
                    dx[(int64_t)ci * S + IDX3(z, yy, xx, H, W)] = (float)acc;
                }
}

void orc_convt_bwd_weight(const float* x, const float* dy, float* dw, float* db, int Cin, int Cout, int D, int H, int W) {
    int Ho = 2 * H, Wo = 2 * W;
    int64_t S = (int64_t)D * H * W, So = 8 * S;
#pragma omp parallel for collapse(2) schedule(static)
    for (int ci = 0; ci < Cin; ++ci)
        for (int co = 0; co < Cout; ++co)
            for (int t = 0; t < 8; ++t) {
                double acc = 0.0;
                for (int z = 0; z < D; ++z)
                    for (int yy = 0; yy < H; ++yy)
                        for (int xx = 0; xx < W; ++xx)
                            acc += (double)x[(int64_t)ci * S + IDX3(z, yy, xx, H, W)] *
                                   (double)dy[(int64_t)co * So +
                                              IDX3(2 * z + (t >> 2), 2 * yy + ((t >> 1) & 1), 2 * xx + (t & 1), Ho, Wo)];
                dw[((int64_t)ci * Cout + co) * 8 + t] += (float)acc;
            }
    if (db)
        for (int co = 0; co < Cout; ++co) {
            double acc = 0.0;
            for (int64_t i = 0; i < So; ++i) acc += dy[(int64_t)co * So + i];
            db[co] += (float)acc;
        }
}

/* ---- InstanceNorm3d(affine) eps 1e-5 (unet.cpp:74-78) and BatchNorm3d(affine, eps=0.0) in train mode
 * with N == 1 (unet.cpp:80-84): per-channel mean and biased variance over D*H*W.  mean/rstd are saved
 * for backward.  running stats (bnorm only, rm != NULL): momentum 0.1, unbiased variance. ---- */
void orc_norm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, int C,
                  int64_t S, double eps, float* rm, float* rv, double momentum) {
#pragma omp parallel for
    for (int c = 0; c < C; ++c) {
        const float* xp = x + (int64_t)c * S;
        double s = 0.0;
        for (int64_t i = 0; i < S; ++i) s += xp[i];
        double m = s / (double)S, q = 0.0;
        for (int64_t i = 0; i < S; ++i) { double d = xp[i] - m; q += d * d; }
        double var = q / (double)S, r = 1.0 / sqrt(var + eps);
        mean[c] = (float)m; rstd[c] = (float)r;
        for (int64_t i = 0; i < S; ++i) y[(int64_t)c * S + i] = (float)(((double)xp[i] - m) * r * gamma[c] + beta[c]);
        if (rm) {
            rm[c] = (float)((1.0 - momentum) * rm[c] + momentum * m);
            rv[c] = (float)((1.0 - momentum) * rv[c] + momentum * (S > 1 ? q / (double)(S - 1) : var));
        }
    }
}

/* BatchNorm3d in eval mode (running stats; after prepare_for_inference mean 0, var 1: unet.cpp:7-22). */
void orc_bnorm_eval_fwd(const float* x, const float* gamma, const float* beta, const float* rm, const float* rv, float* y,
                        int C, int64_t S, double eps) {
#pragma omp parallel for
    for (int c = 0; c < C; ++c) {
        double r = 1.0 / sqrt((double)rv[c] + eps);
        for (int64_t i = 0; i < S; ++i)
            y[(int64_t)c * S + i] = (float)(((double)x[(int64_t)c * S + i] - rm[c]) * r * gamma[c] + beta[c]);
    }
}

void orc_norm_bwd(const float* x, const float* dy, const float* gamma, const float* mean, const float* rstd, float* dx,
                  float* dgamma, float* dbeta, int C, int64_t S) {
#pragma omp parallel for
    for (int c = 0; c < C; ++c) {
        const float *xp = x + (int64_t)c * S, *gp = dy + (int64_t)c * S;
        double m = mean[c], r = rstd[c], s1 = 0.0, s2 = 0.0;
        for (int64_t i = 0; i < S; ++i) { double xh = (xp[i] - m) * r; s1 += gp[i]; s2 += gp[i] * xh; }
        dgamma[c] += (float)s2; dbeta[c] += (float)s1;
        double m1 = s1 / (double)S, m2 = s2 / (double)S;
        for (int64_t i = 0; i < S; ++i) {
            double xh = (xp[i] - m) * r;
            dx[(int64_t)c * S + i] = (float)(gamma[c] * r * ((double)gp[i] - m1 - xh * m2));
        }
    }
}

/* ---- activations, unet.cpp:91-98: 1 relu, 2 leaky_relu(0.01), 3 elu(alpha 1) ---- */
void orc_act_fwd(const float* x, float* y, int64_t n, int kind) {
#pragma omp parallel for
    for (int64_t i = 0; i < n; ++i) {
        float v = x[i];
        y[i] = kind == 1 ? (v > 0 ? v : 0.f) : kind == 2 ? (v > 0 ? v : 0.01f * v) : (v > 0 ? v : (float)expm1((double)v));
    }
}
void orc_act_bwd(const float* x, const float* dy, float* dx, int64_t n, int kind) {
#pragma omp parallel for
    for (int64_t i = 0; i < n; ++i) {
        float v = x[i];
        float d = kind == 1 ? (v > 0 ? 1.f : 0.f) : kind == 2 ? (v > 0 ? 1.f : 0.01f) : (v > 0 ? 1.f : (float)exp((double)v));
        dx[i] = dy[i] * d;
    }
}

/* ---- MaxPool3d(2, stride 2), floor mode (unet.cpp:38-39); first maximum wins ---- */
void orc_maxpool_fwd(const float* x, float* y, int32_t* arg, int C, int D, int H, int W) {
    int Do = D / 2, Ho = H / 2, Wo = W / 2;
#pragma omp parallel for
    for (int c = 0; c < C; ++c)
        for (int z = 0; z < Do; ++z)
            for (int yy = 0; yy < Ho; ++yy)
                for (int xx = 0; xx < Wo; ++xx) {
                    float best = -INFINITY; int32_t bi = (int32_t)IDX3(2 * z, 2 * yy, 2 * xx, H, W);
                    for (int t = 0; t < 8; ++t) {
                        int64_t i = IDX3(2 * z + (t >> 2), 2 * yy + ((t >> 1) & 1), 2 * xx + (t & 1), H, W);
                        float v = x[(int64_t)c * D * H * W + i];
                        if (v > best || isnan(v)) { best = v; bi = (int32_t)i; }
                    }
                    int64_t o = (int64_t)c * Do * Ho * Wo + IDX3(z, yy, xx, Ho, Wo);
                    y[o] = best; arg[o] = bi;
                }
}
void orc_maxpool_bwd(const float* dy, const int32_t* arg, float* dx, int C, int D, int H, int W) {
    int64_t S = (int64_t)D * H * W, So = (int64_t)(D / 2) * (H / 2) * (W / 2);
    memset(dx, 0, sizeof(float) * C * S);
#pragma omp parallel for
    for (int c = 0; c < C; ++c)
        for (int64_t o = 0; o < So; ++o) dx[(int64_t)c * S + arg[(int64_t)c * So + o]] += dy[(int64_t)c * So + o];
}

/* ---- Upsample(scale 2, nearest) (unet.cpp:41-44) ---- */
void orc_upsample_fwd(const float* x, float* y, int C, int D, int H, int W) {
#pragma omp parallel for
    for (int c = 0; c < C; ++c)
        for (int z = 0; z < 2 * D; ++z)
            for (int yy = 0; yy < 2 * H; ++yy)
                for (int xx = 0; xx < 2 * W; ++xx)
                    y[(int64_t)c * 8 * D * H * W + IDX3(z, yy, xx, 2 * H, 2 * W)] =
                        x[(int64_t)c * D * H * W + IDX3(z >> 1, yy >> 1, xx >> 1, H, W)];
}
void orc_upsample_bwd(const float* dy, float* dx, int C, int D, int H, int W) {
#pragma omp parallel for
    for (int c = 0; c < C; ++c)
        for (int z = 0; z < D; ++z)
            for (int yy = 0; yy < H; ++yy)
                for (int xx = 0; xx < W; ++xx) {
                    double acc = 0.0;
                    for (int t = 0; t < 8; ++t)
                        acc += dy[(int64_t)c * 8 * D * H * W +
                                  IDX3(2 * z + (t >> 2), 2 * yy + ((t >> 1) & 1), 2 * xx + (t & 1), 2 * H, 2 * W)];
                    dx[(int64_t)c * D * H * W + IDX3(z, yy, xx, H, W)] = (float)acc;
                }
}

/* ---- deep-supervision target: nearest interpolate to half size through a float round trip
 * (train.cpp:645-662); src = min(floor(dst * in/out), in-1) as torch 'nearest' does ---- */
void orc_target_half(const int64_t* t, int64_t* o, int D, int H, int W) {
    int Do = D >> 1, Ho = H >> 1, Wo = W >> 1;
    float sd = (float)D / Do, sh = (float)H / Ho, sw = (float)W / Wo;
    for (int z = 0; z < Do; ++z)
        for (int y = 0; y < Ho; ++y)
            for (int x = 0; x < Wo; ++x) {
                int iz = (int)floorf(z * sd), iy = (int)floorf(y * sh), ix = (int)floorf(x * sw);
                if (iz > D - 1) iz = D - 1;
                if (iy > H - 1) iy = H - 1;
                if (ix > W - 1) ix = W - 1;
                o[IDX3(z, y, x, Ho, Wo)] = (int64_t)(float)t[IDX3(iz, iy, ix, H, W)];
            }
}

/* ---- calc_losses, train.cpp:501-552, forward and dL/dlogits of  w_ce*ce + w_dice*dice + w_mse*mse.
 * logits [C,S], target [S] (int64; target >= C is "invalid" and masked out, :523-526).
 * collapse_before k (:514-521): classes 0..k-1 merged by logsumexp into class 0.
 * out3 = {ce, dice, mse}; dlogits may be NULL. ---- */
void orc_calc_losses(const float* logits, const int64_t* target, int C, int64_t S, int collapse, double w_ce, double w_dice,
                     double w_mse, double* out3, float* dlogits) {
    int oc = collapse ? C - collapse + 1 : C;
    double* prob = (double*)malloc(sizeof(double) * oc * S);   /* clamped softmax */
    double* sm = (double*)malloc(sizeof(double) * oc * S);     /* raw softmax */
    int32_t* tg = (int32_t*)malloc(sizeof(int32_t) * S);
    uint8_t* vd = (uint8_t*)malloc(S);
    double nvalid = 0.0, ce = 0.0, mse = 0.0;
    double* inter = (double*)calloc(oc, sizeof(double));
    double* card = (double*)calloc(oc, sizeof(double));
    double* lg = (double*)malloc(sizeof(double) * oc);
    for (int64_t i = 0; i < S; ++i) {
        int64_t t = target[i];
        vd[i] = t < C;
        int tt = collapse ? (int)(t - collapse + 1 > 0 ? t - collapse + 1 : 0) : (int)t;
        if (!vd[i]) tt = 0;
        tg[i] = tt;
        if (collapse) {
            double mx = -INFINITY;
            for (int c = 0; c < collapse; ++c) if (logits[(int64_t)c * S + i] > mx) mx = logits[(int64_t)c * S + i];
            double s = 0.0;
            for (int c = 0; c < collapse; ++c) s += exp(logits[(int64_t)c * S + i] - mx);
            lg[0] = mx + log(s);
            for (int c = collapse; c < C; ++c) lg[c - collapse + 1] = logits[(int64_t)c * S + i];
        } else
            for (int c = 0; c < C; ++c) lg[c] = logits[(int64_t)c * S + i];
        double mx = -INFINITY, s = 0.0;
        for (int c = 0; c < oc; ++c) if (lg[c] > mx) mx = lg[c];
        for (int c = 0; c < oc; ++c) s += exp(lg[c] - mx);
        double psq = 0.0;
        for (int c = 0; c < oc; ++c) {
            double p = exp(lg[c] - mx) / s;
            sm[(int64_t)c * S + i] = p;
            double pc = p < 1e-6 ? 1e-6 : (p > 1.0 - 1e-6 ? 1.0 - 1e-6 : p);
            prob[(int64_t)c * S + i] = pc;
            psq += pc * pc;
        }
        if (vd[i]) {
            nvalid += 1.0;
            ce += -(lg[tt] - mx - log(s));
            mse += psq - 2.0 * prob[(int64_t)tt * S + i] + 1.0;
            for (int c = 1; c < oc; ++c) {
                double p = prob[(int64_t)c * S + i], m = (tt == c) ? 1.0 : 0.0;
                inter[c] += p * m; card[c] += p + m;
            }
        }
    }
    double n = nvalid < 1.0 ? 1.0 : nvalid;
    /* eps is a float32 tensor 1e-5 in the reference (train.cpp:538) */
    double eps = (double)1e-5f, dice_sum = 0.0;
    for (int c = 1; c < oc; ++c) dice_sum += (2.0 * inter[c] + eps) / (card[c] + eps);
    int dden = oc - 1 > 1 ? oc - 1 : 1;
    out3[0] = ce / n; out3[1] = 1.0 - dice_sum / dden; out3[2] = mse / n;
    if (dlogits) {
        double* dp = (double*)malloc(sizeof(double) * oc);
        double* dl = (double*)malloc(sizeof(double) * oc);
        for (int64_t i = 0; i < S; ++i) {
            int tt = tg[i];
            double v = vd[i] ? 1.0 : 0.0;
            /* d/dprob (clamped) of mse and dice, then through clamp (grad 1 inside [1e-6,1-1e-6]) and softmax */
            double dot = 0.0;
            for (int c = 0; c < oc; ++c) {
                double p = prob[(int64_t)c * S + i], g = 0.0;
                g += w_mse * v * (2.0 * p - (c == tt ? 2.0 : 0.0)) / n;
                if (c >= 1) {
                    double m = (tt == c) ? 1.0 : 0.0, den = card[c] + eps;
                    g += w_dice * v * (-(2.0 * m * den - (2.0 * inter[c] + eps)) / (den * den)) / dden;
                }
                double q = sm[(int64_t)c * S + i];
                if (!(q >= 1e-6 && q <= 1.0 - 1e-6)) g = 0.0;
                dp[c] = g; dot += g * q;
            }
            for (int c = 0; c < oc; ++c) {
                double q = sm[(int64_t)c * S + i];
                dl[c] = q * (dp[c] - dot) + w_ce * v * (q - (c == tt ? 1.0 : 0.0)) / n;
            }
            if (collapse) {
                double mx = -INFINITY, s = 0.0;
                for (int c = 0; c < collapse; ++c) if (logits[(int64_t)c * S + i] > mx) mx = logits[(int64_t)c * S + i];
                for (int c = 0; c < collapse; ++c) s += exp(logits[(int64_t)c * S + i] - mx);
                for (int c = 0; c < collapse; ++c)
                    dlogits[(int64_t)c * S + i] = (float)(dl[0] * exp(logits[(int64_t)c * S + i] - mx) / s);
                for (int c = collapse; c < C; ++c) dlogits[(int64_t)c * S + i] = (float)dl[c - collapse + 1];
            } else
                for (int c = 0; c < C; ++c) dlogits[(int64_t)c * S + i] = (float)dl[c];
        }
        free(dp); free(dl);
    }
    free(prob); free(sm); free(tg); free(vd); free(inter); free(card); free(lg);
}

/* ---- step epilogue, train.cpp:759-766 + torch::optim::SGD(momentum .99, nesterov, weight decay) as
 * configured at unet.cpp:254-275.  grads already divided by batch_size by the caller.
 * clip_grad_norm_(max 12.0): coef = min(1, max/(norm+1e-6)).  first_step: momentum buffer = grad. ---- */
double orc_grad_norm(const float* g, int64_t n) {
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) s += (double)g[i] * g[i];
    return sqrt(s);
}
void orc_sgd_step(float* p, const float* g, float* mom, int64_t n, double lr, double momentum, double wd, double clip_coef,
                  int first_step) {
    for (int64_t i = 0; i < n; ++i) {
        double d = (double)g[i] * clip_coef + wd * p[i];
        double b = first_step ? d : momentum * mom[i] + d;
        mom[i] = (float)b;
        p[i] = (float)((double)p[i] - lr * (d + momentum * b));
    }
}
