// HBM-bound kernels of the UNet3d path: layout conversion, norm statistics and their backward,
// pooling / resampling, copies, the fused losses of train.cpp:501-552 and the step epilogue of
// train.cpp:759-766.  All reductions are two-stage (per-block partials, then a finalize kernel that
// sums the partials in a fixed order in fp64): deterministic, no float atomics.
#include "device_util.h"

namespace unet {

// ------------------------------------------------------------------------------------------------
// layout: fp32 NCDHW <-> channels-last element type
// ------------------------------------------------------------------------------------------------
template <typename T> __global__ void k_pack_input(const float* __restrict__ x, T* __restrict__ y, int C, int64_t S) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // output index v*C + c
    if (i >= S * C) return;
    int c = (int)(i % C); int64_t v = i / C;
    st<T>(y, i, x[(int64_t)c * S + v]);
}
void launch_pack_input(int dtype, const float* x, void* y, int C, int64_t S, hipStream_t s) {
    UNET_DISPATCH(dtype, (k_pack_input<T><<<cdiv64(S * C, 256), 256, 0, s>>>(x, (T*)y, C, S)));
}

template <typename T> __global__ void k_export(SrcDesc src, float* __restrict__ y, int64_t S) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // output index c*S + v (coalesced writes)
    if (i >= S * src.C) return;
    int c = (int)(i / S); int64_t v = i % S;
    y[i] = view_ld<T>(src, v, c);
}
void launch_export(int dtype, SrcDesc src, float* y, int64_t S, hipStream_t s) {
    UNET_DISPATCH(dtype, (k_export<T><<<cdiv64(S * src.C, 256), 256, 0, s>>>(src, y, S)));
}
void launch_unpack_ncdhw(int dtype, const void* g, float* out, int C, int64_t S, hipStream_t s) {
    SrcDesc d;
    d.ptr = g; d.C = C;
    launch_export(dtype, d, out, S, s);
}

template <typename T> __global__ void k_import_grad(const float* __restrict__ g, T* __restrict__ o, int C, int64_t S, int acc) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= S * C) return;
    int c = (int)(i % C); int64_t v = i / C;
    float val = g[(int64_t)c * S + v];
    if (acc) val += ld<T>(o, i);
    st<T>(o, i, val);
}
void launch_import_grad(int dtype, const float* g, void* o, int C, int64_t S, int accumulate, hipStream_t s) {
    UNET_DISPATCH(dtype, (k_import_grad<T><<<cdiv64(S * C, 256), 256, 0, s>>>(g, (T*)o, C, S, accumulate)));
}
void launch_export_bwd(int dtype, const float* g, DstGrad dst, int64_t S, hipStream_t s) {
    if (!dst.ptr) return;
    launch_import_grad(dtype, g, dst.ptr, dst.C, S, dst.accumulate, s);
}

// ------------------------------------------------------------------------------------------------
// norm statistics.  Block b covers voxels [b*VPB, (b+1)*VPB); threads = (voxel lane) x (channel).
// ------------------------------------------------------------------------------------------------
static inline int64_t stats_vpb(int64_t S) {
    int64_t vpb = (S + 1023) / 1024;
    return vpb < 32 ? 32 : vpb;          // small (deep-level) tensors still get up to S/32 blocks
}
int stats_blocks(int64_t S) {
    int64_t vpb = stats_vpb(S);
    return (int)((S + vpb - 1) / vpb);
}
static inline int pow2_ge(int c) { int p = 1; while (p < c && p < 256) p <<= 1; return p; }

// Accumulator of the statistics kernels: fp32 tensors (the fp32 engine = the parity configuration) are summed in fp64 and leave
// fp64 block partials, as ATen's CPU norm kernels do (acc_type<float> is double there).  The sums cancel -- sum dv*xhat of the
// norm backward is a projection residue, E[x^2]-mean^2 a difference of near-equal numbers when the mean dwarfs the spread --
// and with fp32 sums the weight gradients behind a norm were 3e-3 of the largest gradient away from an fp64 evaluation.
template <typename T> struct StatAcc { typedef float type; };
template <> struct StatAcc<float> { typedef double type; };

// MODE 0: {sum x, sum x^2};  MODE 1 (norm backward): g <- dv = g*act'(v); {sum dv, sum dv*xhat}
// partial: [block][C][2] of StatAcc<T>::type (float for bf16 tensors, double for fp32 tensors)
template <typename T, int MODE>
__global__ void __launch_bounds__(256) k_stats_partial(const T* __restrict__ x, T* __restrict__ gbuf, int C, int64_t S, int64_t VPB,
                                                       int CW, const float* __restrict__ stat, int act, float* __restrict__ partial_) {
    typedef typename StatAcc<T>::type A;
    A* partial = (A*)partial_;
    __shared__ A red[2][256];
    int NV = 256 / CW, cl = threadIdx.x % CW, lane = threadIdx.x / CW;
    int64_t v0 = (int64_t)blockIdx.x * VPB, v1 = v0 + VPB < S ? v0 + VPB : S;
    for (int cg = 0; cg < C; cg += CW) {
        int c = cg + cl;
        A s1 = 0, s2 = 0;
        if (c < C) {
            float mean = 0.f, rstd = 1.f, sc = 1.f, sh = 0.f;
            if (MODE == 1) { mean = stat[c]; rstd = stat[C + c]; sc = stat[2 * C + c]; sh = stat[3 * C + c]; }
            for (int64_t v = v0 + lane; v < v1; v += NV) {
                float u = ld<T>(x, v * C + c);
                if (MODE == 0) { s1 += (A)u; s2 += (A)u * (A)u; }
                else {   // dv is not stored: the apply pass recomputes it from the same inputs
                    float dv = ld<T>(gbuf, v * C + c) * act_d(fmaf(u, sc, sh), act);
                    s1 += (A)dv; s2 += (A)dv * (A)((u - mean) * rstd);
                }
            }
        }
        red[0][threadIdx.x] = s1; red[1][threadIdx.x] = s2;
        __syncthreads();
        if (lane == 0 && c < C) {
            A a = 0, b = 0;
            for (int l = 0; l < NV; ++l) { a += red[0][l * CW + cl]; b += red[1][l * CW + cl]; }
            partial[((int64_t)blockIdx.x * C + c) * 2 + 0] = a;
            partial[((int64_t)blockIdx.x * C + c) * 2 + 1] = b;
        }
        __syncthreads();
    }
}
void launch_stats_partial(int dtype, const void* x, int C, int64_t S, float* partial, hipStream_t s) {
    int nb = stats_blocks(S);
    UNET_DISPATCH(dtype, (k_stats_partial<T, 0><<<nb, 256, 0, s>>>((const T*)x, nullptr, C, S, stats_vpb(S), pow2_ge(C), nullptr, 0, partial)));
}
// bf16, C/8 a power of two <= 256: 16 B (8 channels) per load.  thread = (voxel lane, 8-channel group).
__device__ __forceinline__ void unpack8(uint4 v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__global__ void __launch_bounds__(256) k_norm_bwd_stats8(const uint4* __restrict__ u8, const uint4* __restrict__ g8, int C, int64_t S,
                                                         int64_t VPB, const float* __restrict__ stat, int act, float* __restrict__ partial) {
    __shared__ float red[256][17];
    const int G8 = C / 8, NV = 256 / G8, grp = threadIdx.x % G8, lane = threadIdx.x / G8, c0 = grp * 8;
    const int64_t v0 = (int64_t)blockIdx.x * VPB, v1 = v0 + VPB < S ? v0 + VPB : S;
    float mean[8], rstd[8], sc[8], sh[8], s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        mean[e] = stat[c0 + e]; rstd[e] = stat[C + c0 + e]; sc[e] = stat[2 * C + c0 + e]; sh[e] = stat[3 * C + c0 + e];
        s1[e] = 0.f; s2[e] = 0.f;
    }
    for (int64_t v = v0 + lane; v < v1; v += 2 * NV) {     // two voxels per trip: four 16-B loads in flight per thread
        const bool two = v + NV < v1;
        const uint4 ru0 = u8[v * G8 + grp], rg0 = g8[v * G8 + grp];
        uint4 ru1 = ru0, rg1 = make_uint4(0u, 0u, 0u, 0u);
        if (two) { ru1 = u8[(v + NV) * G8 + grp]; rg1 = g8[(v + NV) * G8 + grp]; }
        float uf[8], gf[8];
        unpack8(ru0, uf);
        unpack8(rg0, gf);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float dv = gf[e] * act_d(fmaf(uf[e], sc[e], sh[e]), act);
            s1[e] += dv; s2[e] = fmaf(dv, (uf[e] - mean[e]) * rstd[e], s2[e]);
        }
        if (two) {
            unpack8(ru1, uf);
            unpack8(rg1, gf);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float dv = gf[e] * act_d(fmaf(uf[e], sc[e], sh[e]), act);
                s1[e] += dv; s2[e] = fmaf(dv, (uf[e] - mean[e]) * rstd[e], s2[e]);
            }
        }
    }
    // threads with equal tid % G8 hold sums of the same 8 channels.  G8 <= 32: shuffle tree over the lanes G8 apart, then the four
    // wave totals through LDS (the serial loop over NV = 256 / G8 LDS rows per output -- 128 at 16 channels -- was as long as the
    // block's whole voxel loop).  G8 = 64..256: a wave holds each group at most once, the (<= 4) rows are summed from LDS as before.
    const int ln = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (G8 <= 32) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float a = s1[e], b = s2[e];
            for (int m = G8; m < 64; m <<= 1) { a += __shfl_xor(a, m); b += __shfl_xor(b, m); }
            s1[e] = a; s2[e] = b;
        }
        if (ln < G8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { red[wv * G8 + ln][e] = s1[e]; red[wv * G8 + ln][8 + e] = s2[e]; }
        }
        __syncthreads();
        if (threadIdx.x < G8 * 8) {
            const int gg = threadIdx.x / 8, e = threadIdx.x % 8;
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { a += red[w * G8 + gg][e]; b += red[w * G8 + gg][8 + e]; }
            partial[((int64_t)blockIdx.x * C + gg * 8 + e) * 2 + 0] = a;
            partial[((int64_t)blockIdx.x * C + gg * 8 + e) * 2 + 1] = b;
        }
        return;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[threadIdx.x][e] = s1[e]; red[threadIdx.x][8 + e] = s2[e]; }
    __syncthreads();
    for (int o = threadIdx.x; o < G8 * 8; o += 256) {
        const int gg = o / 8, e = o % 8;
        float a = 0.f, b = 0.f;
        for (int l = 0; l < NV; ++l) { a += red[l * G8 + gg][e]; b += red[l * G8 + gg][8 + e]; }
        partial[((int64_t)blockIdx.x * C + gg * 8 + e) * 2 + 0] = a;
        partial[((int64_t)blockIdx.x * C + gg * 8 + e) * 2 + 1] = b;
    }
}
// same thread geometry as k_norm_bwd_stats8: a thread keeps its 8 channels' coefficients in registers and walks voxels.
// du = A*da*act'(v) + B*u + D   with  A = coef0, B = -coef0*rstd*m2, D = -coef0*(m1 - mean*rstd*m2)
__global__ void __launch_bounds__(256) k_norm_bwd_apply8(uint4* __restrict__ g8, const uint4* __restrict__ u8, int C, int64_t S, int64_t VPB,
                                                         const float* __restrict__ stat, const float* __restrict__ coef, int act) {
    const int G8 = C / 8, NV = 256 / G8, grp = threadIdx.x % G8, lane = threadIdx.x / G8, c0 = grp * 8;
    const int64_t v0 = (int64_t)blockIdx.x * VPB, v1 = v0 + VPB < S ? v0 + VPB : S;
    float sc[8], sh[8], A[8], B[8], D[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = c0 + e;
        const float mean = stat[c], rstd = stat[C + c];
        sc[e] = stat[2 * C + c]; sh[e] = stat[3 * C + c];
        A[e] = coef[c];
        B[e] = -coef[c] * rstd * coef[2 * C + c];
        D[e] = -coef[c] * (coef[C + c] - mean * rstd * coef[2 * C + c]);
    }
    for (int64_t v = v0 + lane; v < v1; v += NV) {
        float uf[8], gf[8];
        unpack8(u8[v * G8 + grp], uf);
        unpack8(g8[v * G8 + grp], gf);
        unsigned w[4];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            float r0 = fmaf(A[e] * gf[e], act_d(fmaf(uf[e], sc[e], sh[e]), act), fmaf(B[e], uf[e], D[e]));
            float r1 = fmaf(A[e + 1] * gf[e + 1], act_d(fmaf(uf[e + 1], sc[e + 1], sh[e + 1]), act), fmaf(B[e + 1], uf[e + 1], D[e + 1]));
            w[e / 2] = (unsigned)__bfloat16_as_ushort(__float2bfloat16(r0)) | ((unsigned)__bfloat16_as_ushort(__float2bfloat16(r1)) << 16);
        }
        g8[v * G8 + grp] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}
// per-block column sums of a bf16 [S][C] tensor (bias gradient), same thread geometry as k_norm_bwd_stats8
__global__ void __launch_bounds__(256) k_colsum8(const uint4* __restrict__ x8, int C, int64_t S, int64_t VPB, float* __restrict__ partial) {
    __shared__ float red[256][9];
    const int G8 = C / 8, NV = 256 / G8, grp = threadIdx.x % G8, lane = threadIdx.x / G8;
    const int64_t v0 = (int64_t)blockIdx.x * VPB, v1 = v0 + VPB < S ? v0 + VPB : S;
    float s1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = 0.f;
    for (int64_t v = v0 + lane; v < v1; v += NV) {
        float f[8];
        unpack8(x8[v * G8 + grp], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) s1[e] += f[e];
    }
    const int ln = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (G8 <= 32) {      // as in k_norm_bwd_stats8: shuffle tree over the lanes G8 apart, then the four wave totals
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float a = s1[e];
            for (int m = G8; m < 64; m <<= 1) a += __shfl_xor(a, m);
            s1[e] = a;
        }
        if (ln < G8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) red[wv * G8 + ln][e] = s1[e];
        }
        __syncthreads();
        if (threadIdx.x < G8 * 8) {
            const int gg = threadIdx.x / 8, e = threadIdx.x % 8;
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) a += red[w * G8 + gg][e];
            partial[(int64_t)blockIdx.x * C + gg * 8 + e] = a;
        }
        return;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = s1[e];
    __syncthreads();
    for (int o = threadIdx.x; o < G8 * 8; o += 256) {
        const int gg = o / 8, e = o % 8;
        float a = 0.f;
        for (int l = 0; l < NV; ++l) a += red[l * G8 + gg][e];
        partial[(int64_t)blockIdx.x * C + gg * 8 + e] = a;
    }
}
static inline bool vec8_ok(int dtype, int C);
// partial[blk][C]; returns the number of blocks, 0 when the vectorised form does not apply
int launch_colsum_partial8(int dtype, const void* x, int C, int64_t S, float* partial, hipStream_t s) {
    if (!vec8_ok(dtype, C)) return 0;
    int nb = stats_blocks(S);
    k_colsum8<<<nb, 256, 0, s>>>((const uint4*)x, C, S, stats_vpb(S), partial);
    return nb;
}

static inline bool vec8_ok(int dtype, int C) {
    if (dtype != 1 || C % 8) return false;
    int g8 = C / 8;
    return g8 <= 256 && (g8 & (g8 - 1)) == 0;
}
void launch_norm_bwd_partial(int dtype, void* g, const void* u, int C, int64_t S, const float* stat, int act, float* partial,
                             hipStream_t s) {
    int nb = stats_blocks(S);
    if (vec8_ok(dtype, C)) {
        k_norm_bwd_stats8<<<nb, 256, 0, s>>>((const uint4*)u, (const uint4*)g, C, S, stats_vpb(S), stat, act, partial);
        return;
    }
    UNET_DISPATCH(dtype, (k_stats_partial<T, 1><<<nb, 256, 0, s>>>((const T*)u, (T*)g, C, S, stats_vpb(S), pow2_ge(C), stat, act, partial)));
}

// block of 256 threads (4 waves): shuffle tree inside each wave, the four wave sums added in wave order by thread 0
__device__ __forceinline__ void reduce2_wave(double& a, double& b) {
    __shared__ double wred[4][2];
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o); b += __shfl_down(b, o); }
    if ((threadIdx.x & 63) == 0) { wred[threadIdx.x >> 6][0] = a; wred[threadIdx.x >> 6][1] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = wred[0][0]; b = wred[0][1];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { a += wred[w][0]; b += wred[w][1]; }
    }
}

// one block (1 or 4 waves) per channel
// block partials [blk][C][2]: float, or (dbl) double as k_stats_partial leaves them for fp32 tensors
__device__ __forceinline__ void sum_partials(const float* __restrict__ partial, int dbl, int nblk, int C, int c, double& a, double& b) {
    a = 0.0; b = 0.0;
    if (dbl) {
        const double* pd = (const double*)partial;
        for (int i = threadIdx.x; i < nblk; i += blockDim.x) { a += pd[((int64_t)i * C + c) * 2]; b += pd[((int64_t)i * C + c) * 2 + 1]; }
    } else {
        for (int i = threadIdx.x; i < nblk; i += blockDim.x) { a += partial[((int64_t)i * C + c) * 2]; b += partial[((int64_t)i * C + c) * 2 + 1]; }
    }
}

__global__ void __launch_bounds__(256) k_norm_finalize(const float* __restrict__ partial, int nblk, int C, int64_t S,
                                                      const float* gamma, const float* beta, double eps, float* stat, float* rm,
                                                      float* rv, double momentum, int dbl) {
    int c = blockIdx.x;
    double a, b;
    sum_partials(partial, dbl, nblk, C, c, a, b);
    reduce2_wave(a, b);
    if (threadIdx.x == 0) {
        double mean = a / (double)S, var = b / (double)S - mean * mean;
        if (var < 0.0) var = 0.0;
        double rstd = 1.0 / sqrt(var + eps);
        double sc = (double)gamma[c] * rstd;
        stat[c] = (float)mean; stat[C + c] = (float)rstd; stat[2 * C + c] = (float)sc; stat[3 * C + c] = (float)((double)beta[c] - mean * sc);
        if (rm) {
            rm[c] = (float)((1.0 - momentum) * rm[c] + momentum * mean);
            rv[c] = (float)((1.0 - momentum) * rv[c] + momentum * (S > 1 ? var * (double)S / (double)(S - 1) : var));
        }
    }
}
void launch_norm_finalize(const float* partial, int nblk, int C, int64_t S, const float* gamma, const float* beta, double eps,
                          float* stat, float* rm, float* rv, double momentum, hipStream_t s, bool dbl) {
    k_norm_finalize<<<C, nblk > 128 ? 256 : 64, 0, s>>>(partial, nblk, C, S, gamma, beta, eps, stat, rm, rv, momentum, dbl ? 1 : 0);
}

// out[c][j] = sum over blocks of partial[blk][c][j]   (one wave per channel, fp64, fixed order)
__global__ void __launch_bounds__(256) k_stats_sum(const float* __restrict__ partial, int nblk, int C, float* __restrict__ out, int dbl) {
    int c = blockIdx.x;
    double a, b;
    sum_partials(partial, dbl, nblk, C, c, a, b);
    reduce2_wave(a, b);
    if (threadIdx.x == 0) { out[c * 2] = (float)a; out[c * 2 + 1] = (float)b; }
}
void launch_stats_sum(const float* partial, int nblk, int C, float* out, hipStream_t s, bool dbl) {
    k_stats_sum<<<C, nblk > 128 ? 256 : 64, 0, s>>>(partial, nblk, C, out, dbl ? 1 : 0);
}

__global__ void k_norm_eval(int C, const float* gamma, const float* beta, const float* rm, const float* rv, double eps, float* stat) {
    int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    double rstd = 1.0 / sqrt((double)rv[c] + eps), sc = (double)gamma[c] * rstd;
    stat[c] = rm[c]; stat[C + c] = (float)rstd; stat[2 * C + c] = (float)sc; stat[3 * C + c] = (float)((double)beta[c] - (double)rm[c] * sc);
}
void launch_norm_eval(int C, const float* gamma, const float* beta, const float* rm, const float* rv, double eps, float* stat,
                      hipStream_t s) {
    k_norm_eval<<<(C + 63) / 64, 64, 0, s>>>(C, gamma, beta, rm, rv, eps, stat);
}

__global__ void __launch_bounds__(256) k_norm_bwd_finalize(const float* __restrict__ partial, int nblk, int C, int64_t S,
                                                          const float* gamma, const float* stat, float* coef, float* dgamma,
                                                          float* dbeta, int dbl) {
    int c = blockIdx.x;
    double a, b;
    sum_partials(partial, dbl, nblk, C, c, a, b);
    reduce2_wave(a, b);
    if (threadIdx.x == 0) {
        coef[c] = gamma[c] * stat[C + c];
        coef[C + c] = (float)(a / (double)S);
        coef[2 * C + c] = (float)(b / (double)S);
        dgamma[c] += (float)b;
        dbeta[c] += (float)a;
    }
}
void launch_norm_bwd_finalize(const float* partial, int nblk, int C, int64_t S, const float* gamma, const float* stat, float* coef,
                              float* dgamma, float* dbeta, hipStream_t s, bool dbl) {
    k_norm_bwd_finalize<<<C, nblk > 128 ? 256 : 64, 0, s>>>(partial, nblk, C, S, gamma, stat, coef, dgamma, dbeta, dbl ? 1 : 0);
}

// norm backward with few partial rows: k_norm_bwd_finalize in the prologue of k_norm_bwd_apply8 (see k_norm_finalize_apply8)
__device__ __forceinline__ void prologue_sums(const float* __restrict__ partial, int nblk, int C, double* sa, double* sb);
__global__ void __launch_bounds__(256) k_norm_bwd_finalize_apply8(const float* __restrict__ partial, int nblk, int C, int64_t S, int64_t VPB,
                                                                  const float* __restrict__ gamma, const float* __restrict__ stat,
                                                                  float* __restrict__ coef, float* dgamma, float* dbeta,
                                                                  uint4* __restrict__ g8, const uint4* __restrict__ u8, int act) {
    __shared__ double sa[512], sb[512];
    __shared__ float s_c0[512], s_m1[512], s_m2[512];
    prologue_sums(partial, nblk, C, sa, sb);
    for (int c = threadIdx.x; c < C; c += 256) {          // k_norm_bwd_finalize's arithmetic
        s_c0[c] = gamma[c] * stat[C + c];
        s_m1[c] = (float)(sa[c] / (double)S);
        s_m2[c] = (float)(sb[c] / (double)S);
        if (blockIdx.x == 0) {
            coef[c] = s_c0[c]; coef[C + c] = s_m1[c]; coef[2 * C + c] = s_m2[c];
            dgamma[c] += (float)sb[c];
            dbeta[c] += (float)sa[c];
        }
    }
    __syncthreads();
    const int G8 = C / 8, NV = 256 / G8, grp = threadIdx.x % G8, lane = threadIdx.x / G8, c0 = grp * 8;
    const int64_t v0 = (int64_t)blockIdx.x * VPB, v1 = v0 + VPB < S ? v0 + VPB : S;
    float sc[8], sh[8], A[8], B[8], D[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = c0 + e;
        const float mean = stat[c], rstd = stat[C + c];
        sc[e] = stat[2 * C + c]; sh[e] = stat[3 * C + c];
        A[e] = s_c0[c];
        B[e] = -s_c0[c] * rstd * s_m2[c];
        D[e] = -s_c0[c] * (s_m1[c] - mean * rstd * s_m2[c]);
    }
    for (int64_t v = v0 + lane; v < v1; v += NV) {
        float uf[8], gf[8];
        unpack8(u8[v * G8 + grp], uf);
        unpack8(g8[v * G8 + grp], gf);
        unsigned w[4];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            float r0 = fmaf(A[e] * gf[e], act_d(fmaf(uf[e], sc[e], sh[e]), act), fmaf(B[e], uf[e], D[e]));
            float r1 = fmaf(A[e + 1] * gf[e + 1], act_d(fmaf(uf[e + 1], sc[e + 1], sh[e + 1]), act), fmaf(B[e + 1], uf[e + 1], D[e + 1]));
            w[e / 2] = (unsigned)__bfloat16_as_ushort(__float2bfloat16(r0)) | ((unsigned)__bfloat16_as_ushort(__float2bfloat16(r1)) << 16);
        }
        g8[v * G8 + grp] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}
template <typename T> __global__ void k_norm_bwd_apply(T* __restrict__ g, const T* __restrict__ u, int C, int64_t n,
                                                       const float* __restrict__ stat, const float* __restrict__ coef, int act) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int c = (int)(i % C);
    float uu = ld<T>(u, i);
    float dv = ld<T>(g, i) * act_d(fmaf(uu, stat[2 * C + c], stat[3 * C + c]), act);
    float xh = (uu - stat[c]) * stat[C + c];
    st<T>(g, i, coef[c] * (dv - coef[C + c] - xh * coef[2 * C + c]));
}
void launch_norm_bwd_apply(int dtype, void* g, const void* u, int C, int64_t S, const float* stat, const float* coef, int act,
                           hipStream_t s) {
    if (vec8_ok(dtype, C)) {
        int64_t vpb = (S + 4095) / 4096;          // up to 4096 blocks, each a contiguous voxel range
        if (vpb < 32) vpb = 32;
        k_norm_bwd_apply8<<<cdiv64(S, vpb), 256, 0, s>>>((uint4*)g, (const uint4*)u, C, S, vpb, stat, coef, act);
        return;
    }
    UNET_DISPATCH(dtype, (k_norm_bwd_apply<T><<<cdiv64(S * C, 256), 256, 0, s>>>((T*)g, (const T*)u, C, S * C, stat, coef, act)));
}

static inline bool fused_norm_ok(int dtype, int C, int nblk);
static inline int64_t fused_vpb(int C, int64_t S);
bool launch_norm_bwd_finalize_apply(int dtype, const float* partial, int nblk, int C, int64_t S, const float* gamma, const float* stat,
                                    float* coef, float* dgamma, float* dbeta, void* g, const void* u, int act, hipStream_t s) {
    if (!fused_norm_ok(dtype, C, nblk)) return false;
    const int64_t vpb = fused_vpb(C, S);
    k_norm_bwd_finalize_apply8<<<cdiv64(S, vpb), 256, 0, s>>>(partial, nblk, C, S, vpb, gamma, stat, coef, dgamma, dbeta, (uint4*)g,
                                                              (const uint4*)u, act);
    return true;
}

template <typename T> __global__ void k_act_bwd(T* __restrict__ g, const T* __restrict__ u, int act, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    st<T>(g, i, ld<T>(g, i) * act_d(ld<T>(u, i), act));
}
void launch_act_bwd(int dtype, void* g, const void* u, int act, int64_t n, hipStream_t s) {
    UNET_DISPATCH(dtype, (k_act_bwd<T><<<cdiv64(n, 256), 256, 0, s>>>((T*)g, (const T*)u, act, n)));
}

// ------------------------------------------------------------------------------------------------
// MaxPool3d(2,2) floor mode (unet.cpp:38-39), Upsample nearest x2 (unet.cpp:41-44), copies
// ------------------------------------------------------------------------------------------------
template <typename T> __global__ void k_maxpool_fwd(SrcDesc src, T* __restrict__ out, int D, int H, int W) {
    int Do = D / 2, Ho = H / 2, Wo = W / 2, C = src.C;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)Do * Ho * Wo * C) return;
    int c = (int)(i % C); int64_t v = i / C;
    int x = (int)(v % Wo); int64_t r = v / Wo;
    int y = (int)(r % Ho); int z = (int)(r / Ho);
    float best = -INFINITY;
    for (int t = 0; t < 8; ++t) {
        int64_t vi = ((int64_t)(2 * z + (t >> 2)) * H + (2 * y + ((t >> 1) & 1))) * W + (2 * x + (t & 1));
        float val = view_ld<T>(src, vi, c);
        if (val > best || val != val) best = val;
    }
    st<T>(out, i, best);
}
void launch_maxpool_fwd(int dtype, SrcDesc src, void* out, int D, int H, int W, hipStream_t s) {
    int64_t n = (int64_t)(D / 2) * (H / 2) * (W / 2) * src.C;
    UNET_DISPATCH(dtype, (k_maxpool_fwd<T><<<cdiv64(n, 256), 256, 0, s>>>(src, (T*)out, D, H, W)));
}
// one thread per INPUT element: gradient goes to the first maximum of its window (windows do not overlap)
template <typename T> __global__ void k_maxpool_bwd(SrcDesc src, const T* __restrict__ gout, DstGrad dst, int D, int H, int W) {
    int Do = D / 2, Ho = H / 2, Wo = W / 2, C = src.C;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)D * H * W * C) return;
    int c = (int)(i % C); int64_t v = i / C;
    int x = (int)(v % W); int64_t r = v / W;
    int y = (int)(r % H); int z = (int)(r / H);
    float val = 0.f;
    int zo = z >> 1, yo = y >> 1, xo = x >> 1;
    if (zo < Do && yo < Ho && xo < Wo) {
        float best = -INFINITY; int bt = 0;
        for (int t = 0; t < 8; ++t) {
            int64_t vi = ((int64_t)(2 * zo + (t >> 2)) * H + (2 * yo + ((t >> 1) & 1))) * W + (2 * xo + (t & 1));
            float a = view_ld<T>(src, vi, c);
            if (a > best || a != a) { best = a; bt = t; }
        }
        int mine = ((z & 1) << 2) | ((y & 1) << 1) | (x & 1);
        if (mine == bt) val = ld<T>(gout, (((int64_t)zo * Ho + yo) * Wo + xo) * C + c);
    }
    T* p = (T*)dst.ptr;
    if (dst.accumulate) val += ld<T>(p, i);
    st<T>(p, i, val);
}
void launch_maxpool_bwd(int dtype, SrcDesc src, const void* gout, DstGrad dst, int D, int H, int W, hipStream_t s) {
    if (!dst.ptr) return;
    int64_t n = (int64_t)D * H * W * src.C;
    UNET_DISPATCH(dtype, (k_maxpool_bwd<T><<<cdiv64(n, 256), 256, 0, s>>>(src, (const T*)gout, dst, D, H, W)));
}

template <typename T> __global__ void k_upsample_fwd(SrcDesc src, T* __restrict__ out, int D, int H, int W) {
    int C = src.C;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)8 * D * H * W * C) return;
    int c = (int)(i % C); int64_t v = i / C;
    int x = (int)(v % (2 * W)); int64_t r = v / (2 * W);
    int y = (int)(r % (2 * H)); int z = (int)(r / (2 * H));
    st<T>(out, i, view_ld<T>(src, ((int64_t)(z >> 1) * H + (y >> 1)) * W + (x >> 1), c));
}
void launch_upsample_fwd(int dtype, SrcDesc src, void* out, int D, int H, int W, hipStream_t s) {
    int64_t n = (int64_t)8 * D * H * W * src.C;
    UNET_DISPATCH(dtype, (k_upsample_fwd<T><<<cdiv64(n, 256), 256, 0, s>>>(src, (T*)out, D, H, W)));
}
template <typename T> __global__ void k_upsample_bwd(const T* __restrict__ gout, DstGrad dst, int D, int H, int W) {
    int C = dst.C;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)D * H * W * C) return;
    int c = (int)(i % C); int64_t v = i / C;
    int x = (int)(v % W); int64_t r = v / W;
    int y = (int)(r % H); int z = (int)(r / H);
    float val = 0.f;
    for (int t = 0; t < 8; ++t)
        val += ld<T>(gout, (((int64_t)(2 * z + (t >> 2)) * (2 * H) + (2 * y + ((t >> 1) & 1))) * (2 * W) + (2 * x + (t & 1))) * C + c);
    T* p = (T*)dst.ptr;
    if (dst.accumulate) val += ld<T>(p, i);
    st<T>(p, i, val);
}
void launch_upsample_bwd(int dtype, const void* gout, DstGrad dst, int D, int H, int W, hipStream_t s) {
    if (!dst.ptr) return;
    int64_t n = (int64_t)D * H * W * dst.C;
    UNET_DISPATCH(dtype, (k_upsample_bwd<T><<<cdiv64(n, 256), 256, 0, s>>>((const T*)gout, dst, D, H, W)));
}

struct Src2 { SrcDesc s[2]; int n; };
struct Dst2 { DstGrad d[2]; int n; };
template <typename T> __global__ void k_materialize(Src2 src, T* __restrict__ out, int C, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int c = (int)(i % C); int64_t v = i / C;
    float val = (src.n > 1 && c >= src.s[0].C) ? view_ld<T>(src.s[1], v, c - src.s[0].C) : view_ld<T>(src.s[0], v, c);
    st<T>(out, i, val);
}
// act(x*scale+shift) of a whole bf16 tensor, 8 channels (16 B) per thread: the activated copy consumers read
__global__ void __launch_bounds__(256) k_apply_view8(SrcDesc src, uint4* __restrict__ out, int64_t n8) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    int c0 = (int)((i * 8) % src.C);
    uint4 v = ((const uint4*)src.ptr)[i];
    unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float lo = __uint_as_float(w[e] << 16), hi = __uint_as_float(w[e] & 0xffff0000u);
        if (src.scale) {
            lo = fmaf(lo, src.scale[c0 + 2 * e], src.shift[c0 + 2 * e]);
            hi = fmaf(hi, src.scale[c0 + 2 * e + 1], src.shift[c0 + 2 * e + 1]);
        }
        lo = act_f(lo, src.act); hi = act_f(hi, src.act);
        w[e] = (unsigned)__bfloat16_as_ushort(__float2bfloat16(lo)) | ((unsigned)__bfloat16_as_ushort(__float2bfloat16(hi)) << 16);
    }
    out[i] = make_uint4(w[0], w[1], w[2], w[3]);
}
// the same copy in the geometry of k_norm_bwd_apply8 (a thread keeps its 8 channels' scale / shift in registers and walks voxels,
// four 16-B loads in flight): 134 MB in 32 us (one unit per thread, 16 coefficient loads each) -> see profiles/tools/elem_bench.cpp
__global__ void __launch_bounds__(256) k_apply_view8g(SrcDesc src, uint4* __restrict__ out, int64_t S, int64_t VPB) {
    const int G8 = src.C / 8, NV = 256 / G8, grp = threadIdx.x % G8, lane = threadIdx.x / G8, c0 = grp * 8;
    const int64_t v0 = (int64_t)blockIdx.x * VPB, v1 = v0 + VPB < S ? v0 + VPB : S;
    const uint4* in = (const uint4*)src.ptr;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = src.scale ? src.scale[c0 + e] : 1.f; sh[e] = src.scale ? src.shift[c0 + e] : 0.f; }
    const int act = src.act;
    auto one = [&](uint4 v) {
        unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float lo = act_f(fmaf(__uint_as_float(w[e] << 16), sc[2 * e], sh[2 * e]), act);
            float hi = act_f(fmaf(__uint_as_float(w[e] & 0xffff0000u), sc[2 * e + 1], sh[2 * e + 1]), act);
            w[e] = (unsigned)__bfloat16_as_ushort(__float2bfloat16(lo)) | ((unsigned)__bfloat16_as_ushort(__float2bfloat16(hi)) << 16);
        }
        return make_uint4(w[0], w[1], w[2], w[3]);
    };
    int64_t v = v0 + lane;
    for (; v + 3 * NV < v1; v += 4 * NV) {
        uint4 r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = in[(v + j * NV) * G8 + grp];
#pragma unroll
        for (int j = 0; j < 4; ++j) out[(v + j * NV) * G8 + grp] = one(r[j]);
    }
    for (; v < v1; v += NV) out[v * G8 + grp] = one(in[v * G8 + grp]);
}
void launch_apply_view(int dtype, SrcDesc src, void* out, int64_t S, hipStream_t s) {
    if (dtype == 1 && src.C % 8 == 0 && src.C <= 2048 && 256 % (src.C / 8) == 0 && S * src.C >= (1 << 18)) {
        int64_t vpb = (S + 4095) / 4096;
        const int64_t nv4 = 4 * (256 / (src.C / 8));
        if (vpb < nv4) vpb = nv4;
        vpb = (vpb + nv4 - 1) / nv4 * nv4;           // whole trips of four units per thread
        k_apply_view8g<<<cdiv64(S, vpb), 256, 0, s>>>(src, (uint4*)out, S, vpb);
        return;
    }
    if (dtype == 1 && src.C % 8 == 0) {
        int64_t n8 = S * src.C / 8;
        k_apply_view8<<<cdiv64(n8, 256), 256, 0, s>>>(src, (uint4*)out, n8);
    } else {
        launch_materialize(dtype, &src, 1, out, S, s);
    }
}

// ---- few partial rows (the 32^3 and deeper levels): the finalize moves into the prologue of the consumer ----
// A kernel boundary costs ~5 us whatever the kernel does, and at these levels k_norm_finalize / k_norm_bwd_finalize ARE that cost (22 + 22
// launches of 4.7 us per step).  With <= 128 partial rows every block of the element-wise pass can re-sum them itself: C x rows x 8 B
// from L2 (<= 64 KB), one thread per (channel, row quarter), fp64, fixed order; block 0 also leaves the values the later passes read
// (stat / coef, running statistics, dgamma / dbeta).  (For the 64^3 and 128^3 levels -- 512..1024 rows -- that re-read is as much L2
// traffic as the tensor itself, measured slower: they keep the separate finalize.)
constexpr int FUSED_MAX_ROWS = 128, FUSED_MAX_C = 512;
__device__ __forceinline__ void prologue_sums(const float* __restrict__ partial, int nblk, int C, double* sa, double* sb) {
    // sa / sb: LDS, 512 doubles each.  A thread owns a channel PAIR (one 16-B load per row: {a, b} of two channels) and every RG-th row;
    // its loads are issued sixteen at a time before the first add (one load per trip of a rolled loop is one exposed L2 latency per
    // row: the first version of this prologue took longer than the launch it saved); the RG row groups of a channel are then added in order.
    const int CP = C / 2, RG = CP >= 256 ? 1 : 256 / CP;
    for (int idx = threadIdx.x; idx < CP * RG; idx += 256) {
        const int cp = idx % CP, rg = idx / CP;
        double a0 = 0.0, b0 = 0.0, a1 = 0.0, b1 = 0.0;
        for (int r0 = rg; r0 < nblk; r0 += 16 * RG) {
            float4 v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int r = r0 + k * RG;
                v[k] = r < nblk ? *(const float4*)(partial + ((int64_t)r * C + 2 * cp) * 2) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) { a0 += v[k].x; b0 += v[k].y; a1 += v[k].z; b1 += v[k].w; }
        }
        sa[rg * C + 2 * cp] = a0; sb[rg * C + 2 * cp] = b0; sa[rg * C + 2 * cp + 1] = a1; sb[rg * C + 2 * cp + 1] = b1;
    }
    __syncthreads();
    if (RG > 1) {
        for (int c = threadIdx.x; c < C; c += 256) {         // C <= 256 here: one pass; a thread reads its channel's RG sums, then overwrites slot 0
            double a = 0.0, b = 0.0;
            for (int q = 0; q < RG; ++q) { a += sa[q * C + c]; b += sb[q * C + c]; }
            sa[c] = a; sb[c] = b;                            // slot (0, c) is only read by this thread
        }
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256) k_norm_finalize_apply8(const float* __restrict__ partial, int nblk, int C, int64_t S, int64_t VPB,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, double eps,
                                                              float* __restrict__ stat, float* rm, float* rv, double momentum,
                                                              const uint4* __restrict__ in, int act, uint4* __restrict__ out) {
    __shared__ double sa[FUSED_MAX_C], sb[FUSED_MAX_C];
    __shared__ float s_sc[FUSED_MAX_C], s_sh[FUSED_MAX_C];
    prologue_sums(partial, nblk, C, sa, sb);
    for (int c = threadIdx.x; c < C; c += 256) {          // k_norm_finalize's arithmetic
        const double mean = sa[c] / (double)S;
        double var = sb[c] / (double)S - mean * mean;
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(var + eps), sc = (double)gamma[c] * rstd;
        s_sc[c] = (float)sc; s_sh[c] = (float)((double)beta[c] - mean * sc);
        if (blockIdx.x == 0) {
            stat[c] = (float)mean; stat[C + c] = (float)rstd; stat[2 * C + c] = s_sc[c]; stat[3 * C + c] = s_sh[c];
            if (rm) {
                rm[c] = (float)((1.0 - momentum) * rm[c] + momentum * mean);
                rv[c] = (float)((1.0 - momentum) * rv[c] + momentum * (S > 1 ? var * (double)S / (double)(S - 1) : var));
            }
        }
    }
    __syncthreads();
    const int G8 = C / 8, NV = 256 / G8, grp = threadIdx.x % G8, lane = threadIdx.x / G8, c0 = grp * 8;
    const int64_t v0 = (int64_t)blockIdx.x * VPB, v1 = v0 + VPB < S ? v0 + VPB : S;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = s_sc[c0 + e]; sh[e] = s_sh[c0 + e]; }
    for (int64_t v = v0 + lane; v < v1; v += NV) {
        const uint4 r = in[v * G8 + grp];
        unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float lo = act_f(fmaf(__uint_as_float(w[e] << 16), sc[2 * e], sh[2 * e]), act);
            float hi = act_f(fmaf(__uint_as_float(w[e] & 0xffff0000u), sc[2 * e + 1], sh[2 * e + 1]), act);
            w[e] = (unsigned)__bfloat16_as_ushort(__float2bfloat16(lo)) | ((unsigned)__bfloat16_as_ushort(__float2bfloat16(hi)) << 16);
        }
        out[v * G8 + grp] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}
static inline bool fused_norm_ok(int dtype, int C, int nblk) {
    return dtype == 1 && C % 8 == 0 && C <= FUSED_MAX_C && 256 % (C / 8) == 0 && nblk <= FUSED_MAX_ROWS;
}
static inline int64_t fused_vpb(int C, int64_t S) {
    const int64_t nv = 256 / (C / 8);
    int64_t vpb = (S + 255) / 256;                     // <= 256 blocks: every block repeats the prologue
    if (vpb < 2 * nv) vpb = 2 * nv;
    return (vpb + nv - 1) / nv * nv;
}
bool launch_norm_finalize_apply(int dtype, const float* partial, int nblk, int C, int64_t S, const float* gamma, const float* beta, double eps,
                                float* stat, float* rm, float* rv, double momentum, const void* raw, int act, void* out, hipStream_t s) {
    if (!fused_norm_ok(dtype, C, nblk)) return false;
    const int64_t vpb = fused_vpb(C, S);
    k_norm_finalize_apply8<<<cdiv64(S, vpb), 256, 0, s>>>(partial, nblk, C, S, vpb, gamma, beta, eps, stat, rm, rv, momentum,
                                                          (const uint4*)raw, act, (uint4*)out);
    return true;
}

void launch_materialize(int dtype, const SrcDesc* src, int nsrc, void* out, int64_t S, hipStream_t s) {
    Src2 a;
    a.n = nsrc; a.s[0] = src[0]; if (nsrc > 1) a.s[1] = src[1];
    int C = src[0].C + (nsrc > 1 ? src[1].C : 0);
    UNET_DISPATCH(dtype, (k_materialize<T><<<cdiv64(S * C, 256), 256, 0, s>>>(a, (T*)out, C, S * C)));
}
template <typename T> __global__ void k_materialize_bwd(const T* __restrict__ gout, Dst2 dst, int C, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int c = (int)(i % C); int64_t v = i / C;
    int which = (dst.n > 1 && c >= dst.d[0].C) ? 1 : 0;
    const DstGrad& d = dst.d[which];
    if (!d.ptr) return;
    int64_t j = v * d.C + (which ? c - dst.d[0].C : c);
    float val = ld<T>(gout, i);
    T* p = (T*)d.ptr;
    if (d.accumulate) val += ld<T>(p, j);
    st<T>(p, j, val);
}
void launch_materialize_bwd(int dtype, const void* gout, const DstGrad* dst, int ndst, int64_t S, hipStream_t s) {
    Dst2 a;
    a.n = ndst; a.d[0] = dst[0]; if (ndst > 1) a.d[1] = dst[1];
    int C = dst[0].C + (ndst > 1 ? dst[1].C : 0);
    UNET_DISPATCH(dtype, (k_materialize_bwd<T><<<cdiv64(S * C, 256), 256, 0, s>>>((const T*)gout, a, C, S * C)));
}

// ------------------------------------------------------------------------------------------------
// (exp / log are the hardware v_exp_f32 / v_log_f32 forms (__expf, __logf, ~2 ulp): the loss kernels were bound by the precise
// library versions, 18 exps per voxel; the 1e-4 loss parity and the gradient checks against the oracle hold, tests/test_gpu_parity.py)
// losses: calc_losses (train.cpp:501-552) over fp32 NCDHW logits and int64 targets
// ------------------------------------------------------------------------------------------------
// deep-supervision target pyramid, train.cpp:645-662: nearest, src = min(floor(dst * in/out), in-1)
__global__ void k_target_half(const int64_t* __restrict__ t, int64_t* __restrict__ o, int D, int H, int W) {
    int Do = D >> 1, Ho = H >> 1, Wo = W >> 1;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)Do * Ho * Wo) return;
    int x = (int)(i % Wo); int64_t r = i / Wo;
    int y = (int)(r % Ho); int z = (int)(r / Ho);
    float sd = (float)D / Do, sh = (float)H / Ho, sw = (float)W / Wo;
    int iz = min((int)floorf(z * sd), D - 1), iy = min((int)floorf(y * sh), H - 1), ix = min((int)floorf(x * sw), W - 1);
    o[i] = (int64_t)(float)t[((int64_t)iz * H + iy) * W + ix];
}
void launch_target_half(const int64_t* t, int64_t* o, int D, int H, int W, hipStream_t s) {
    int64_t n = (int64_t)(D >> 1) * (H >> 1) * (W >> 1);
    k_target_half<<<cdiv64(n, 256), 256, 0, s>>>(t, o, D, H, W);
}

int loss_blocks(int64_t S) {
    int64_t nb = (S + 255) / 256;
    return (int)(nb > 1024 ? 1024 : nb);
}

// merged logit of class c' under collapse_before k (train.cpp:514-521): c' = 0 is logsumexp of classes 0..k-1
struct VoxelLogits {
    const float* p; int64_t S, v; int k; float lse0;
    __device__ float get(int cp) const { return (k && cp == 0) ? lse0 : p[(int64_t)(k ? cp + k - 1 : cp) * S + v]; }
};
__device__ __forceinline__ VoxelLogits make_vl(const float* logits, int64_t S, int64_t v, int k) {
    VoxelLogits L;
    L.p = logits; L.S = S; L.v = v; L.k = k; L.lse0 = 0.f;
    if (k) {
        float mx = -INFINITY;
        for (int c = 0; c < k; ++c) mx = fmaxf(mx, logits[(int64_t)c * S + v]);
        float s = 0.f;
        for (int c = 0; c < k; ++c) s += __expf(logits[(int64_t)c * S + v] - mx);
        L.lse0 = mx + __logf(s);
    }
    return L;
}
__device__ __forceinline__ float clamp_p(float q) { return fminf(fmaxf(q, 1e-6f), 1.0f - 1e-6f); }

// partial layout per block: [0] ce, [1] mse, [2] nvalid, [3 .. 3+oc) inter, [3+oc .. 3+2oc) card
__global__ void __launch_bounds__(256) k_loss_partial(const float* __restrict__ logits, const int64_t* __restrict__ target, int C,
                                                      int64_t S, int k, float* __restrict__ partial) {
    extern __shared__ float sh[];  // 3 + 2*oc
    int oc = k ? C - k + 1 : C;
    int np = 3 + 2 * oc;
    for (int i = threadIdx.x; i < np; i += 256) sh[i] = 0.f;
    __syncthreads();
    float ce = 0.f, mse = 0.f, nv = 0.f;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < S; v += (int64_t)gridDim.x * 256) {
        int64_t t = target[v];
        bool valid = t < C;
        if (!valid) continue;  // invalid voxels carry weight 0 in every term (train.cpp:523-550)
        int tt = k ? (int)(t - k + 1 > 0 ? t - k + 1 : 0) : (int)t;
        VoxelLogits L = make_vl(logits, S, v, k);
        float mx = -INFINITY;
        for (int c = 0; c < oc; ++c) mx = fmaxf(mx, L.get(c));
        float sum = 0.f;
        for (int c = 0; c < oc; ++c) sum += __expf(L.get(c) - mx);
        float inv = 1.f / sum, psq = 0.f, pt = 0.f;
        for (int c = 0; c < oc; ++c) {
            float p = clamp_p(__expf(L.get(c) - mx) * inv);
            psq = fmaf(p, p, psq);
            if (c == tt) pt = p;
            if (c >= 1) atomicAdd(&sh[3 + oc + c], p + (c == tt ? 1.f : 0.f));
        }
        if (tt >= 1) atomicAdd(&sh[3 + tt], pt);
        ce += -(L.get(tt) - mx - __logf(sum));
        mse += psq - 2.f * pt + 1.f;
        nv += 1.f;
    }
    atomicAdd(&sh[0], ce); atomicAdd(&sh[1], mse); atomicAdd(&sh[2], nv);
    __syncthreads();
    for (int i = threadIdx.x; i < np; i += 256) partial[(int64_t)blockIdx.x * np + i] = sh[i];
}
// Same partials with per-thread register accumulators and a fixed-order block reduction: bit-exact from run to run.
// Used whenever the (merged) class count fits OCMAX; the LDS-atomic kernel above only serves larger class counts.
template <int OCMAX>
__global__ void __launch_bounds__(256) k_loss_partial_reg(const float* __restrict__ logits, const int64_t* __restrict__ target, int C,
                                                          int64_t S, int k, float* __restrict__ partial) {
    constexpr int RS = 3 + 2 * OCMAX + 1;   // LDS row per wave
    __shared__ float red[4 * RS];
    const int oc = k ? C - k + 1 : C, np = 3 + 2 * oc;
    float ce = 0.f, mse = 0.f, nv = 0.f, inter[OCMAX], card[OCMAX];
#pragma unroll
    for (int c = 0; c < OCMAX; ++c) { inter[c] = 0.f; card[c] = 0.f; }
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < S; v += (int64_t)gridDim.x * 256) {
        int64_t t = target[v];
        if (!(t < C)) continue;
        int tt = k ? (int)(t - k + 1 > 0 ? t - k + 1 : 0) : (int)t;
        VoxelLogits L = make_vl(logits, S, v, k);
        float lg[OCMAX], mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < OCMAX; ++c) { lg[c] = c < oc ? L.get(c) : -INFINITY; mx = fmaxf(mx, lg[c]); }
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < OCMAX; ++c) { lg[c] = c < oc ? __expf(lg[c] - mx) : 0.f; sum += lg[c]; }
        float inv = 1.f / sum, psq = 0.f, pt = 0.f, lt = 0.f;
#pragma unroll
        for (int c = 0; c < OCMAX; ++c) {
            if (c < oc) {
                float p = clamp_p(lg[c] * inv);
                psq = fmaf(p, p, psq);
                bool hit = c == tt;
                if (hit) { pt = p; lt = L.get(c); inter[c] += p; }
                card[c] += p + (hit ? 1.f : 0.f);
            }
        }
        ce += -(lt - mx - __logf(sum));
        mse += psq - 2.f * pt + 1.f;
        nv += 1.f;
    }
    // block sums: xor-shuffles inside a wave, then the 4 wave results through LDS (one barrier; the np block-wide tree reductions
    // of the first version were 120 barriers per block -- the whole cost of the kernel on the small levels)
    auto wsum = [](float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; };
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float w3[3] = {wsum(ce), wsum(mse), wsum(nv)};
    if (lane == 0) { red[wave * RS + 0] = w3[0]; red[wave * RS + 1] = w3[1]; red[wave * RS + 2] = w3[2]; }
#pragma unroll
    for (int c = 0; c < OCMAX; ++c) {
        if (c < oc) {
            const float a = wsum(inter[c]), b = wsum(card[c]);
            if (lane == 0) { red[wave * RS + 3 + c] = a; red[wave * RS + 3 + oc + c] = b; }
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < np)
        partial[(int64_t)blockIdx.x * np + threadIdx.x] = red[threadIdx.x] + red[RS + threadIdx.x] + red[2 * RS + threadIdx.x] + red[3 * RS + threadIdx.x];
}
// no class merging, <= 8 classes, S % 4 == 0 (every level of the default training set-up): four consecutive voxels per thread and
// trip -- one 16-B load per class plane and two for the int64 targets instead of 4-B / 8-B ones (67 MB in 22 us -> see
// profiles/tools/elem_bench.cpp); the same per-thread register accumulators and fixed-order block reduction
template <int OCMAX>
__global__ void __launch_bounds__(256) k_loss_partial_v4(const float* __restrict__ logits, const int64_t* __restrict__ target, int C,
                                                         int64_t S, float* __restrict__ partial) {
    constexpr int RS = 3 + 2 * OCMAX + 1;
    __shared__ float red[4 * RS];
    const int oc = C, np = 3 + 2 * oc;
    float ce = 0.f, mse = 0.f, nv = 0.f, inter[OCMAX], card[OCMAX];
#pragma unroll
    for (int c = 0; c < OCMAX; ++c) { inter[c] = 0.f; card[c] = 0.f; }
    const int64_t S4 = S >> 2;
    for (int64_t q4 = (int64_t)blockIdx.x * 256 + threadIdx.x; q4 < S4; q4 += (int64_t)gridDim.x * 256) {
        const longlong2 t01 = ((const longlong2*)target)[2 * q4], t23 = ((const longlong2*)target)[2 * q4 + 1];
        const int64_t tv[4] = {t01.x, t01.y, t23.x, t23.y};
        float4 pl[OCMAX];
#pragma unroll
        for (int c = 0; c < OCMAX; ++c)
            pl[c] = c < oc ? ((const float4*)(logits + (int64_t)c * S))[q4] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t t = tv[j];
            if (!(t < C)) continue;
            const int tt = (int)t;
            float lg[OCMAX], mx = -INFINITY;
#pragma unroll
            for (int c = 0; c < OCMAX; ++c) { lg[c] = j == 0 ? pl[c].x : j == 1 ? pl[c].y : j == 2 ? pl[c].z : pl[c].w; mx = fmaxf(mx, lg[c]); }
            float sum = 0.f, lt = 0.f;
#pragma unroll
            for (int c = 0; c < OCMAX; ++c) { if (c == tt) lt = lg[c]; lg[c] = c < oc ? __expf(lg[c] - mx) : 0.f; sum += lg[c]; }
            const float inv = 1.f / sum;
            float psq = 0.f, pt = 0.f;
#pragma unroll
            for (int c = 0; c < OCMAX; ++c) {
                if (c < oc) {
                    const float p = clamp_p(lg[c] * inv);
                    psq = fmaf(p, p, psq);
                    const bool hit = c == tt;
                    if (hit) { pt = p; inter[c] += p; }
                    card[c] += p + (hit ? 1.f : 0.f);
                }
            }
            ce += -(lt - mx - __logf(sum));
            mse += psq - 2.f * pt + 1.f;
            nv += 1.f;
        }
    }
    auto wsum = [](float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; };
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float w3[3] = {wsum(ce), wsum(mse), wsum(nv)};
    if (lane == 0) { red[wave * RS + 0] = w3[0]; red[wave * RS + 1] = w3[1]; red[wave * RS + 2] = w3[2]; }
#pragma unroll
    for (int c = 0; c < OCMAX; ++c) {
        if (c < oc) {
            const float a = wsum(inter[c]), b = wsum(card[c]);
            if (lane == 0) { red[wave * RS + 3 + c] = a; red[wave * RS + 3 + oc + c] = b; }
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < np)
        partial[(int64_t)blockIdx.x * np + threadIdx.x] = red[threadIdx.x] + red[RS + threadIdx.x] + red[2 * RS + threadIdx.x] + red[3 * RS + threadIdx.x];
}
static inline bool loss_v4_ok(const void* a, const void* b, const void* c, int C, int64_t S, int collapse) {
    return !collapse && C <= 8 && S % 4 == 0 && S >= 4096 && ((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) % 16 == 0;
}
void launch_loss_partial(const float* logits, const int64_t* target, int C, int64_t S, int collapse, float* partial, hipStream_t s) {
    int oc = collapse ? C - collapse + 1 : C;
    if (loss_v4_ok(logits, target, nullptr, C, S, collapse)) {
        // loss_blocks(S) rows are summed by the finalize: every one of them is written (S/4 >= 256 * blocks for S >= 2^20; smaller
        // levels: blocks without a voxel write zeros)
        k_loss_partial_v4<8><<<loss_blocks(S), 256, 0, s>>>(logits, target, C, S, partial);
        return;
    }
    if (oc <= 8) k_loss_partial_reg<8><<<loss_blocks(S), 256, 0, s>>>(logits, target, C, S, collapse, partial);
    else if (oc <= 32) k_loss_partial_reg<32><<<loss_blocks(S), 256, 0, s>>>(logits, target, C, S, collapse, partial);
    else k_loss_partial<<<loss_blocks(S), 256, (3 + 2 * oc) * sizeof(float), s>>>(logits, target, C, S, collapse, partial);
}

// level_out: [0] ce [1] dice [2] mse [3] n  [4 .. 4+oc) inter  [4+oc .. 4+2oc) card
// totals: [0] total loss (accumulated over levels) [1..3] level-0 ce, dice, mse
__global__ void __launch_bounds__(1024) k_loss_finalize(const float* __restrict__ partial, int nblk, int oc, float weight, int cost_mask,
                                                       float* level_out, float* totals, int set_stats) {
    extern __shared__ double shd[];  // 3 + 2*oc
    int np = 3 + 2 * oc;
    // one wave per value, 64 row lanes (<= 16 rows each at 1024 blocks), shuffle tree: fixed order.  (16 values x 16 split lanes with
    // 64 dependent 4-byte loads per thread took 18 us at the two largest levels.)
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    for (int i = wv; i < np; i += (int)(blockDim.x >> 6)) {
        double a = 0.0;
        for (int b = ln; b < nblk; b += 64) a += partial[(int64_t)b * np + i];
        for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o);
        if (ln == 0) shd[i] = a;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double n = shd[2] < 1.0 ? 1.0 : shd[2];
        double eps = (double)1e-5f, dsum = 0.0;
        for (int c = 1; c < oc; ++c) dsum += (2.0 * shd[3 + c] + eps) / (shd[3 + oc + c] + eps);
        double ce = shd[0] / n, mse = shd[1] / n, dice = 1.0 - dsum / (double)(oc - 1 > 1 ? oc - 1 : 1);
        level_out[0] = (float)ce; level_out[1] = (float)dice; level_out[2] = (float)mse; level_out[3] = (float)n;
        for (int c = 0; c < oc; ++c) { level_out[4 + c] = (float)shd[3 + c]; level_out[4 + oc + c] = (float)shd[3 + oc + c]; }
        double sel = 0.0;
        if (!(cost_mask & 7)) sel = ce;
        else sel = ((cost_mask & 1) ? ce : 0.0) + ((cost_mask & 2) ? dice : 0.0) + ((cost_mask & 4) ? mse : 0.0);
        totals[0] += (float)(weight * sel);
        if (set_stats) { totals[1] = (float)ce; totals[2] = (float)dice; totals[3] = (float)mse; }
    }
}
void launch_loss_finalize(const float* partial, int nblk, int oc, float weight, int cost_mask, float* level_out, float* totals,
                          int set_stats, hipStream_t s) {
    k_loss_finalize<<<1, 3 + 2 * oc > 8 ? 1024 : 512, (3 + 2 * oc) * sizeof(double), s>>>(partial, nblk, oc, weight, cost_mask, level_out, totals, set_stats);
}

__global__ void __launch_bounds__(256) k_loss_grad(const float* __restrict__ logits, const int64_t* __restrict__ target, int C, int64_t S,
                                                   int k, const float* __restrict__ level_out, float weight, int cost_mask,
                                                   float* __restrict__ dlogits) {
    int oc = k ? C - k + 1 : C;
    int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= S) return;
    int64_t t = target[v];
    if (!(t < C)) {
        for (int c = 0; c < C; ++c) dlogits[(int64_t)c * S + v] = 0.f;
        return;
    }
    float w_ce = (!(cost_mask & 7) || (cost_mask & 1)) ? weight : 0.f;
    float w_dice = (cost_mask & 2) ? weight : 0.f, w_mse = (cost_mask & 4) ? weight : 0.f;
    float n = level_out[3];
    const float* inter = level_out + 4;
    const float* card = level_out + 4 + oc;
    float dden = (float)(oc - 1 > 1 ? oc - 1 : 1), eps = 1e-5f;
    int tt = k ? (int)(t - k + 1 > 0 ? t - k + 1 : 0) : (int)t;
    VoxelLogits L = make_vl(logits, S, v, k);
    float mx = -INFINITY;
    for (int c = 0; c < oc; ++c) mx = fmaxf(mx, L.get(c));
    float sum = 0.f;
    for (int c = 0; c < oc; ++c) sum += __expf(L.get(c) - mx);
    float inv = 1.f / sum;
    // pass A: dot = sum_c dprob_c * q_c
    float dot = 0.f;
    for (int c = 0; c < oc; ++c) {
        float q = __expf(L.get(c) - mx) * inv, p = clamp_p(q), g = 0.f;
        g += w_mse * (2.f * p - (c == tt ? 2.f : 0.f)) / n;
        if (c >= 1) {
            float m = c == tt ? 1.f : 0.f, den = card[c] + eps;
            g += w_dice * (-(2.f * m * den - (2.f * inter[c] + eps)) / (den * den)) / dden;
        }
        if (!(q >= 1e-6f && q <= 1.0f - 1e-6f)) g = 0.f;  // clamp has zero gradient outside its range
        dot = fmaf(g, q, dot);
    }
    // pass B: dlogit_c = q_c*(dprob_c - dot) + w_ce*(q_c - [c==t])/n
    float d0 = 0.f;
    for (int c = 0; c < oc; ++c) {
        float q = __expf(L.get(c) - mx) * inv, p = clamp_p(q), g = 0.f;
        g += w_mse * (2.f * p - (c == tt ? 2.f : 0.f)) / n;
        if (c >= 1) {
            float m = c == tt ? 1.f : 0.f, den = card[c] + eps;
            g += w_dice * (-(2.f * m * den - (2.f * inter[c] + eps)) / (den * den)) / dden;
        }
        if (!(q >= 1e-6f && q <= 1.0f - 1e-6f)) g = 0.f;
        float dl = q * (g - dot) + w_ce * (q - (c == tt ? 1.f : 0.f)) / n;
        if (k && c == 0) d0 = dl;
        else dlogits[(int64_t)(k ? c + k - 1 : c) * S + v] = dl;
    }
    if (k) {
        float m0 = -INFINITY;
        for (int c = 0; c < k; ++c) m0 = fmaxf(m0, logits[(int64_t)c * S + v]);
        float s0 = 0.f;
        for (int c = 0; c < k; ++c) s0 += __expf(logits[(int64_t)c * S + v] - m0);
        for (int c = 0; c < k; ++c) dlogits[(int64_t)c * S + v] = d0 * __expf(logits[(int64_t)c * S + v] - m0) / s0;
    }
}
// four consecutive voxels per thread (see k_loss_partial_v4): 16-B loads and stores per class plane, the per-class dice terms
// computed once per thread; per voxel the same expressions in the same order as k_loss_grad (bit-identical gradients)
template <int OCMAX>
__global__ void __launch_bounds__(256) k_loss_grad_v4(const float* __restrict__ logits, const int64_t* __restrict__ target, int C, int64_t S,
                                                      const float* __restrict__ level_out, float weight, int cost_mask,
                                                      float* __restrict__ dlogits) {
    const int oc = C;
    const int64_t q4 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q4 >= (S >> 2)) return;
    const float w_ce = (!(cost_mask & 7) || (cost_mask & 1)) ? weight : 0.f;
    const float w_dice = (cost_mask & 2) ? weight : 0.f, w_mse = (cost_mask & 4) ? weight : 0.f;
    const float n = level_out[3];
    const float dden = (float)(oc - 1 > 1 ? oc - 1 : 1), eps = 1e-5f;
    float dice_hit[OCMAX], dice_miss[OCMAX];
#pragma unroll
    for (int c = 0; c < OCMAX; ++c) {
        dice_hit[c] = dice_miss[c] = 0.f;
        if (c >= 1 && c < oc) {
            const float den = level_out[4 + oc + c] + eps, in2 = 2.f * level_out[4 + c] + eps;
            dice_hit[c] = w_dice * (-(2.f * 1.f * den - in2) / (den * den)) / dden;
            dice_miss[c] = w_dice * (-(2.f * 0.f * den - in2) / (den * den)) / dden;
        }
    }
    const longlong2 t01 = ((const longlong2*)target)[2 * q4], t23 = ((const longlong2*)target)[2 * q4 + 1];
    const int64_t tv[4] = {t01.x, t01.y, t23.x, t23.y};
    float4 pl[OCMAX];
#pragma unroll
    for (int c = 0; c < OCMAX; ++c)
        pl[c] = c < oc ? ((const float4*)(logits + (int64_t)c * S))[q4] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    float out[OCMAX][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t t = tv[j];
        const bool valid = t < C;
        const int tt = (int)t;
        float lg[OCMAX], mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < OCMAX; ++c) { lg[c] = j == 0 ? pl[c].x : j == 1 ? pl[c].y : j == 2 ? pl[c].z : pl[c].w; mx = fmaxf(mx, lg[c]); }
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < OCMAX; ++c) { lg[c] = c < oc ? __expf(lg[c] - mx) : 0.f; sum += lg[c]; }
        const float inv = 1.f / sum;
        float g[OCMAX], dot = 0.f;
#pragma unroll
        for (int c = 0; c < OCMAX; ++c) {
            g[c] = 0.f;
            if (c < oc) {
                const float q = lg[c] * inv, p = clamp_p(q);
                float gg = 0.f;
                gg += w_mse * (2.f * p - (c == tt ? 2.f : 0.f)) / n;
                if (c >= 1) gg += c == tt ? dice_hit[c] : dice_miss[c];
                if (!(q >= 1e-6f && q <= 1.0f - 1e-6f)) gg = 0.f;
                dot = fmaf(gg, q, dot);
                g[c] = gg; lg[c] = q;
            }
        }
#pragma unroll
        for (int c = 0; c < OCMAX; ++c)
            out[c][j] = (c < oc && valid) ? lg[c] * (g[c] - dot) + w_ce * (lg[c] - (c == tt ? 1.f : 0.f)) / n : 0.f;
    }
#pragma unroll
    for (int c = 0; c < OCMAX; ++c)
        if (c < oc) ((float4*)(dlogits + (int64_t)c * S))[q4] = make_float4(out[c][0], out[c][1], out[c][2], out[c][3]);
}
void launch_loss_grad(const float* logits, const int64_t* target, int C, int64_t S, int collapse, const float* level_out, float weight,
                      int cost_mask, float* dlogits, hipStream_t s) {
    if (loss_v4_ok(logits, target, dlogits, C, S, collapse)) {
        k_loss_grad_v4<8><<<cdiv64(S >> 2, 256), 256, 0, s>>>(logits, target, C, S, level_out, weight, cost_mask, dlogits);
        return;
    }
    k_loss_grad<<<cdiv64(S, 256), 256, 0, s>>>(logits, target, C, S, collapse, level_out, weight, cost_mask, dlogits);
}

// ------------------------------------------------------------------------------------------------
// out[i] = ((b0[i] + b1[i]) + b2[i]) + ...  (fp32, exactly this association: what ONE buffer holds after the micro-steps have
// accumulated into it one after the other, since 0 + x == x); optionally clears the inputs for the next optimizer step.
// The micro-steps of one optimizer step that ran side by side on two streams each wrote a gradient buffer of their own.
// ------------------------------------------------------------------------------------------------
struct SumBufs { const float* in[UNET_SUM_MAX_BUFFERS]; int n; };
__global__ void __launch_bounds__(256) k_sum_buffers(SumBufs b, float* __restrict__ out, int64_t count, int zero_inputs) {
    for (int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i0 < count; i0 += (int64_t)gridDim.x * 1024) {
        if (i0 + 3 < count) {
            float4 acc = *(const float4*)(b.in[0] + i0);
            for (int k = 1; k < b.n; ++k) {
                const float4 v = *(const float4*)(b.in[k] + i0);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            *(float4*)(out + i0) = acc;
            if (zero_inputs)
                for (int k = 0; k < b.n; ++k)
                    if (b.in[k] != out) *(float4*)(const_cast<float*>(b.in[k]) + i0) = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            for (int j = 0; j < 4 && i0 + j < count; ++j) {
                float acc = b.in[0][i0 + j];
                for (int k = 1; k < b.n; ++k) acc += b.in[k][i0 + j];
                out[i0 + j] = acc;
                if (zero_inputs)
                    for (int k = 0; k < b.n; ++k)
                        if (b.in[k] != out) const_cast<float*>(b.in[k])[i0 + j] = 0.f;
            }
        }
    }
}
void launch_sum_buffers(const float* const* bufs, int n, float* out, int64_t count, int zero_inputs, hipStream_t s) {
    SumBufs b;
    b.n = n;
    for (int k = 0; k < UNET_SUM_MAX_BUFFERS; ++k) b.in[k] = k < n ? bufs[k] : nullptr;
    int64_t nb = (count + 1023) / 1024;
    if (nb > 4096) nb = 4096;
    k_sum_buffers<<<(unsigned)nb, 256, 0, s>>>(b, out, count, zero_inputs);
}

// ------------------------------------------------------------------------------------------------
// step epilogue (train.cpp:759-766, unet.cpp:254-275)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_sumsq_partial(const float* __restrict__ g, int64_t n, float scale, float* __restrict__ partial) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i0 < n; i0 += (int64_t)gridDim.x * 1024) {
        float r = 0.f;
        if (i0 + 3 < n) {
            const float4 v = *(const float4*)(g + i0);
            const float a0 = v.x * scale, a1 = v.y * scale, a2 = v.z * scale, a3 = v.w * scale;
            r = fmaf(a0, a0, fmaf(a1, a1, fmaf(a2, a2, a3 * a3)));
        } else {
            for (int j = 0; j < 4; ++j)
                if (i0 + j < n) { float x = g[i0 + j] * scale; r = fmaf(x, x, r); }
        }
        acc += (double)r;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = (float)red[0];
}
void launch_sumsq_partial(const float* g, int64_t n, float scale, float* partial, int nblk, hipStream_t s) {
    k_sumsq_partial<<<nblk, 256, 0, s>>>(g, n, scale, partial);
}

// One fused pass: total gradient norm from the partials, clip coefficient, weight decay (per parameter segment), Nesterov
// momentum, parameter update and zero_grad.  Four consecutive elements per thread (16-B accesses); the segment table and the
// partials are reduced through LDS once per block.
constexpr int SGD_MAXSEG = 1024;
__global__ void __launch_bounds__(256) k_sgd(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, int64_t n,
                                             const SgdSeg* __restrict__ segs, int nseg, const float* __restrict__ partial, int nblk,
                                             float lr, float momentum, int nesterov, float wdecay, float clip_norm, float grad_scale, float* norm_out) {
    __shared__ double red[256];
    __shared__ float s_coef;
    __shared__ int64_t s_off[SGD_MAXSEG];
    __shared__ float s_wd[SGD_MAXSEG];
    double part = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) part += partial[b];
    red[threadIdx.x] = part;
    const int ns = nseg < SGD_MAXSEG ? nseg : SGD_MAXSEG;
    for (int k = threadIdx.x; k < ns; k += 256) { s_off[k] = segs[k].offset; s_wd[k] = segs[k].wd; }
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float norm = (float)sqrt(red[0]);
        float coef = clip_norm / (norm + 1e-6f);
        s_coef = coef > 1.f ? 1.f : coef;
        if (blockIdx.x == 0 && norm_out) *norm_out = norm;
    }
    __syncthreads();
    const float coef = s_coef * grad_scale;
    auto seg_of = [&](int64_t i) {          // last segment with offset <= i
        int lo = 0, hi = nseg - 1;
        while (lo < hi) {
            int mid = (lo + hi + 1) >> 1;
            int64_t off = mid < SGD_MAXSEG ? s_off[mid] : segs[mid].offset;
            if (off <= i) lo = mid; else hi = mid - 1;
        }
        return lo;
    };
    auto wd_of = [&](int sg) { return (sg < SGD_MAXSEG ? s_wd[sg] : segs[sg].wd) * wdecay; };
    auto upd = [&](float& pv, float& gv, float& mv, float wd) {
        float d = fmaf(wd, pv, gv * coef);
        float b = fmaf(momentum, mv, d);
        mv = b;
        pv = pv - lr * (nesterov ? fmaf(momentum, b, d) : b);
        gv = 0.f;
    };
    const int64_t n4 = n >> 2;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += (int64_t)gridDim.x * 256) {
        const int64_t i = q * 4;
        const int sg = seg_of(i);
        const int64_t next = sg + 1 < nseg ? (sg + 1 < SGD_MAXSEG ? s_off[sg + 1] : segs[sg + 1].offset) : n;
        float4 pv = *(float4*)(p + i), gv = *(float4*)(g + i), mv = *(float4*)(m + i);
        float w0 = wd_of(sg), w1 = w0, w2 = w0, w3 = w0;
        if (next < i + 4) { w1 = wd_of(seg_of(i + 1)); w2 = wd_of(seg_of(i + 2)); w3 = wd_of(seg_of(i + 3)); }
        upd(pv.x, gv.x, mv.x, w0); upd(pv.y, gv.y, mv.y, w1); upd(pv.z, gv.z, mv.z, w2); upd(pv.w, gv.w, mv.w, w3);
        *(float4*)(p + i) = pv; *(float4*)(m + i) = mv; *(float4*)(g + i) = gv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {   // tail
        const int64_t i = (n4 << 2) + threadIdx.x;
        float pv = p[i], gv = g[i], mv = m[i];
        upd(pv, gv, mv, wd_of(seg_of(i)));
        p[i] = pv; m[i] = mv; g[i] = gv;
    }
}
void launch_sgd(float* p, float* g, float* m, int64_t n, const SgdSeg* segs, int nseg, const float* partial, int nblk, float lr,
                float momentum, int nesterov, float wdecay, float clip_norm, float grad_scale, float* norm_out, hipStream_t s) {
    int64_t nb = (n / 4 + 255) / 256;
    if (nb < 1) nb = 1;
    if (nb > 2048) nb = 2048;
    k_sgd<<<(unsigned)nb, 256, 0, s>>>(p, g, m, n, segs, nseg, partial, nblk, lr, momentum, nesterov, wdecay, clip_norm, grad_scale, norm_out);
}

}  // namespace unet
