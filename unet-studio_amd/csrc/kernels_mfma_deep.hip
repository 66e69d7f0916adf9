// The deep levels (8^3 voxels and fewer in the default architecture): split-K implicit GEMM with the norm layer in the epilogue.
//
// At 8^3 / 4^3 a conv is a (64..512 voxels) x (128..512 rows) x (K = taps * Cin up to 13824) GEMM whose only large operand is the
// filter (1.8 .. 7 MB of bf16).  The halo-tile kernels (k_mfma_conv_small, k_mfma_conv_p) give such a layer 16..128 blocks: each
// streams up to 0.4 MB of filter through one CU behind a chain of exposed latencies (12-25 us per conv, profiles/r18_step_launch_sequence.txt)
// and the norm that follows is a second launch (5-9 us, most of it the launch boundary).  Here
//   * the contraction is split over K as well: block = (64 voxels, NTW row tiles, one K range), its four waves take a quarter of the
//     range each; 256..1024 blocks stream the filter once, from every CU;
//   * both MFMA operands come straight from global memory in fragment order: the filter from the existing pack
//     ([chunk][tap][row tile][lane][8], kernels_mfma_conv.hip), the input as 16-B loads of 8 channels of voxel S*out + tap - PAD (the
//     whole level's activations are 32..256 KB: L2-resident, no LDS staging, no barrier in the K loop), PF k-steps ahead in registers;
//   * the four waves' accumulators are summed through LDS; with one K range per output tile the block finishes the tile itself,
//     else it leaves an fp32 partial tile in scratch and takes a ticket (atomic counter): the block that takes the LAST ticket of a
//     tile -- nobody waits for anybody -- sums the partials in K-range order (the result does not depend on arrival order) and runs the
//     epilogue.  Partials are written and read with agent-scope accesses (the 8 XCDs' L2s are not coherent with each other for
//     ordinary ones; see deep_publish);
//   * DEEP_FWD_NORM: the ticket is per ROW TILE (16 output channels) over every voxel tile and K range, so the last arriver holds a
//     channel's whole volume: it writes the raw conv output, takes the norm statistics of the values as stored, finalises them
//     (k_norm_finalize's arithmetic: fp64 mean / variance, running statistics), and writes the activated copy -- conv + norm +
//     activation in ONE launch instead of conv, (statistics in its epilogue,) k_norm_finalize_apply8;
//   * DEEP_BWD_NORM (dgrad whose destination is a norm layer's view and which is the last writer of that gradient): the last arriver
//     adds the old gradient if it accumulates, forms dv = dL/d(view) * act'(..), the sums {dv, dv * xhat} over the volume, the norm's
//     affine gradients, and overwrites the gradient with dL/d(raw) -- dgrad + k_norm_bwd_stats8 + k_norm_bwd_finalize_apply8 in one launch.
// The kinds are the halo-tile kernels' (kernels_mfma_conv.hip header): <S, KD, PAD, SC> = <1,3,1,-> conv s1 forward / dgrad,
// <2,3,1,-> conv s2 forward, <1,1,0,SC> conv_trans forward, <2,2,0,-> conv_trans dgrad, <1,2,0,SC> conv s2 dgrad; same packs, same
// arguments, same rounding points (bias added in fp32, one rounding to bf16, statistics of the rounded values).
// Reference: the layers of unet.cpp:46-98 at the deep levels, their autograd backward (train.cpp:660).
#include "mfma_util.h"

namespace unet {

enum { DEEP_PLAIN = 0, DEEP_FWD_NORM = 1, DEEP_BWD_NORM = 2 };

struct DeepArgs {
    MfmaConvArgs c;      // geometry, sources, packed filter, bias, destinations: as for the halo-tile kernels
    float* part;         // [ksplit][row tile][Vpad][16] fp32 partial tiles
    int* cnt;            // tickets; zero between launches (the last arriver resets its counter)
    int nchunk, ksplit, kper;   // 32-channel chunks; K ranges; k-steps per range (k = tap * nchunk + chunk)
    int V, Vpad, MG;     // voxels of the grid the tiles cover, padded to 64, groups of 64
    DeepNormFwd nf;
    DeepNormBwd nb;
};

// ---- the plain epilogue of one lane: 4 consecutive rows of one voxel ----
template <bool SC>
__device__ __forceinline__ void deep_store(const DeepArgs& a, int m, int row0, f32x4 v) {
    const ConvGeom& g = a.c.g;
    if (m >= a.V) return;
    int c = row0, gx = m % g.Wo, r = m / g.Wo, gy = r % g.Ho, gz = r / g.Ho;
    if (SC) {
        const int tap = c / a.c.sc_C;
        c -= tap * a.c.sc_C;
        gz = 2 * gz + (tap >> 2); gy = 2 * gy + ((tap >> 1) & 1); gx = 2 * gx + (tap & 1);
        if (gz >= a.c.oD || gy >= a.c.oH || gx >= a.c.oW) return;
    }
    const int d = (a.c.nout > 1 && c >= a.c.outC[0]) ? 1 : 0, cd = c - (d ? a.c.outC[0] : 0);
    char* obase = (char*)(d ? a.c.out[1] : a.c.out[0]);
    if (!obase) return;
    const int oC = d ? a.c.outC[1] : a.c.outC[0], oacc = d ? a.c.out_acc[1] : a.c.out_acc[0];
    float v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3];
    if (a.c.bias) { v0 += a.c.bias[c]; v1 += a.c.bias[c + 1]; v2 += a.c.bias[c + 2]; v3 += a.c.bias[c + 3]; }
    uint2* p = (uint2*)(obase + ((((size_t)gz * a.c.oH + gy) * a.c.oW + gx) * oC + cd) * 2);
    if (oacc) {
        const uint2 old = *p;
        v0 += bf_lo(old.x); v1 += bf_hi(old.x); v2 += bf_lo(old.y); v3 += bf_hi(old.y);
    }
    uint2 o;
    o.x = pack_bf16x2(v0, v1); o.y = pack_bf16x2(v2, v3);
    *p = o;
}

// Partial tiles cross XCDs: the 8 L2s are not coherent with each other for ordinary accesses, and the fences that make them so
// (buffer_wbl2 / buffer_inv at agent scope, what __threadfence() emits) write back / invalidate a whole L2 -- with 2000 waves doing that
// beside the side stream's kernels a 512-block launch took 87 us.  Instead the partials themselves are written and read with
// agent-scope accesses (sc1: write-through to / read from the memory side), 8 bytes at a time; nothing else needs to be coherent.
__device__ __forceinline__ void deep_publish(float* p, f32x4 v) {
    unsigned long long lo = (unsigned long long)__float_as_uint(v[0]) | ((unsigned long long)__float_as_uint(v[1]) << 32);
    unsigned long long hi = (unsigned long long)__float_as_uint(v[2]) | ((unsigned long long)__float_as_uint(v[3]) << 32);
    __hip_atomic_store((unsigned long long*)p, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store((unsigned long long*)p + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool COH>     // COH: the partials were written by other blocks of THIS launch (ticket path); else by an earlier launch
__device__ __forceinline__ f32x4 deep_fetch(const float* p) {
    if (!COH) return *(const f32x4*)p;
    const unsigned long long lo = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long hi = __hip_atomic_load((const unsigned long long*)p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return f32x4{__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)), __uint_as_float((unsigned)hi), __uint_as_float((unsigned)(hi >> 32))};
}

// the K ranges of a thread's rows (row it = voxel it * 64 + tid / 4, it < iters <= IT), summed in K-range order; 16 partial rows in
// flight per round whatever the volume: IT rows x 16 / IT K ranges
template <int IT, bool COH>
__device__ __forceinline__ void deep_sum_parts(const float* pp, size_t kstride, int ksplit, int iters, f32x4 (&sum)[8]) {
    constexpr int KU = 16 / IT;
    for (int ks0 = 0; ks0 < ksplit; ks0 += KU) {
        f32x4 t[IT][KU];
#pragma unroll
        for (int it = 0; it < IT; ++it)
#pragma unroll
            for (int u = 0; u < KU; ++u)
                t[it][u] = (it < iters && ks0 + u < ksplit) ? deep_fetch<COH>(pp + (size_t)(ks0 + u) * kstride + (size_t)it * 64 * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int it = 0; it < IT; ++it)
#pragma unroll
            for (int u = 0; u < KU; ++u) { sum[it][0] += t[it][u][0]; sum[it][1] += t[it][u][1]; sum[it][2] += t[it][u][2]; sum[it][3] += t[it][u][3]; }
    }
}

__device__ __attribute__((aligned(16))) unsigned g_deep_zero[4] = {0u, 0u, 0u, 0u};   // what lanes outside the volume read

// sums of 8 per-thread values over the block's threads that share tid & 3 (the same 4 channels), in a fixed order: shuffle tree over
// the wave's lanes, then the four waves in wave order.  out[rq * 4 + r] (sum a), out[16 + rq * 4 + r] (sum b) in fp64.
__device__ __forceinline__ void deep_reduce16(const float (&sa)[4], const float (&sb)[4], double* out /* LDS, 4 * 32 + 32 doubles */) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double* wsum = out + 32;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double u = (double)sa[r], v = (double)sb[r];
#pragma unroll
        for (int msk = 4; msk < 64; msk <<= 1) { u += __shfl_xor(u, msk); v += __shfl_xor(v, msk); }
        if (lane < 4) { wsum[wave * 32 + lane * 4 + r] = u; wsum[wave * 32 + 16 + lane * 4 + r] = v; }
    }
    __syncthreads();
    if (tid < 32) out[tid] = wsum[tid] + wsum[32 + tid] + wsum[64 + tid] + wsum[96 + tid];
    __syncthreads();
}

// The finish of an output tile set by the last arriver: sums the K ranges' partial tiles in K-range order and runs the epilogue (COH:
// the loads come from the memory side).  Loads are requested in batches before the first use (a loop of load -> wait -> add per K range and per row made the 512-voxel levels
// 31-44 us per launch: profiles/r20a_step_launch_sequence.txt).
template <bool SC, int EPI, int NTW, bool COH>
__device__ __forceinline__ void deep_finish(const DeepArgs& a, int mg, int ng, double* dsum, float* s_par) {
    const ConvGeom& g = a.c.g;
    const int tid = threadIdx.x, NTT = g.Cout / 16;
    const int ml = tid >> 2, rq = tid & 3;
    if (EPI == DEEP_PLAIN) {
        const int m = mg * 64 + ml;
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
            const int nt = ng * NTW + n;
            const float* pp = a.part + (((size_t)nt * a.Vpad + m) * 16 + rq * 4);
            const size_t kstride = (size_t)NTT * a.Vpad * 16;
            f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int ks0 = 0; ks0 < a.ksplit; ks0 += 8) {
                f32x4 t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = ks0 + u < a.ksplit ? deep_fetch<COH>(pp + (size_t)(ks0 + u) * kstride) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 8; ++u) { sum[0] += t[u][0]; sum[1] += t[u][1]; sum[2] += t[u][2]; sum[3] += t[u][3]; }
            }
            deep_store<SC>(a, m, nt * 16 + rq * 4, sum);
        }
        return;
    }

    // ---- a norm epilogue: this block holds NTW row tiles (16 channels each) over the whole volume (<= DEEP_MAXIT * 64 voxels) ----
    constexpr int MAXIT = (int)(DEEP_MAX_VOXELS / 64);
    const int C = g.Cout, iters = a.Vpad >> 6;
    for (int n = 0; n < NTW; ++n) {
        const int nt = ng * NTW + n, c = nt * 16 + rq * 4;
        float sa[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
        // the K ranges of this thread's rows, summed in K-range order
        f32x4 sum[MAXIT];
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) sum[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            const float* pp = a.part + (((size_t)nt * a.Vpad + ml) * 16 + rq * 4);
            const size_t kstride = (size_t)NTT * a.Vpad * 16;
            if (iters <= 1) deep_sum_parts<1, COH>(pp, kstride, a.ksplit, iters, sum);
            else if (iters <= 2) deep_sum_parts<2, COH>(pp, kstride, a.ksplit, iters, sum);
            else if (iters <= 4) deep_sum_parts<4, COH>(pp, kstride, a.ksplit, iters, sum);
            else deep_sum_parts<8, COH>(pp, kstride, a.ksplit, iters, sum);
        }
        if (EPI == DEEP_FWD_NORM) {
            char* y = (char*)a.c.out[0];
            float b4[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.c.bias) {
#pragma unroll
                for (int r = 0; r < 4; ++r) b4[r] = a.c.bias[c + r];
            }
            // + bias, round, store the raw output; statistics of the values as stored (kept in registers for the second pass)
            uint2 yb[MAXIT];
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
                const int m = it * 64 + ml;
                yb[it] = make_uint2(0u, 0u);
                if (it < iters && m < a.V) {
                    uint2 ob;
                    ob.x = pack_bf16x2(sum[it][0] + b4[0], sum[it][1] + b4[1]); ob.y = pack_bf16x2(sum[it][2] + b4[2], sum[it][3] + b4[3]);
                    *(uint2*)(y + ((size_t)m * C + c) * 2) = ob;
                    yb[it] = ob;
                    const float r0 = bf_lo(ob.x), r1 = bf_hi(ob.x), r2 = bf_lo(ob.y), r3 = bf_hi(ob.y);
                    sa[0] += r0; sa[1] += r1; sa[2] += r2; sa[3] += r3;
                    sb[0] = fmaf(r0, r0, sb[0]); sb[1] = fmaf(r1, r1, sb[1]); sb[2] = fmaf(r2, r2, sb[2]); sb[3] = fmaf(r3, r3, sb[3]);
                }
            }
            if (!a.nf.use_running) deep_reduce16(sa, sb, dsum);
            if (tid < 16) {                                  // k_norm_finalize's arithmetic (k_norm_eval's with running statistics)
                const int cc = nt * 16 + tid;
                double mean, var, rstd;
                if (a.nf.use_running) { mean = (double)a.nf.rm[cc]; var = (double)a.nf.rv[cc]; rstd = 1.0 / sqrt(var + a.nf.eps); }
                else {
                    mean = dsum[tid] / (double)a.V;
                    var = dsum[16 + tid] / (double)a.V - mean * mean;
                    if (var < 0.0) var = 0.0;
                    rstd = 1.0 / sqrt(var + a.nf.eps);
                }
                const double sc = (double)a.nf.gamma[cc] * rstd;
                const float fsc = (float)sc, fsh = a.nf.use_running ? (float)((double)a.nf.beta[cc] - (double)a.nf.rm[cc] * sc)
                                                                    : (float)((double)a.nf.beta[cc] - mean * sc);
                a.nf.stat[cc] = (float)mean; a.nf.stat[C + cc] = (float)rstd; a.nf.stat[2 * C + cc] = fsc; a.nf.stat[3 * C + cc] = fsh;
                if (a.nf.rm && !a.nf.use_running) {
                    a.nf.rm[cc] = (float)((1.0 - a.nf.momentum) * a.nf.rm[cc] + a.nf.momentum * mean);
                    a.nf.rv[cc] = (float)((1.0 - a.nf.momentum) * a.nf.rv[cc] + a.nf.momentum * (a.V > 1 ? var * (double)a.V / (double)(a.V - 1) : var));
                }
                s_par[tid] = fsc; s_par[16 + tid] = fsh;
            }
            __syncthreads();
            // the activated copy
            float sc4[4], sh4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { sc4[r] = s_par[rq * 4 + r]; sh4[r] = s_par[16 + rq * 4 + r]; }
            char* ao = (char*)a.nf.act_out;
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
                const int m = it * 64 + ml;
                if (it < iters && m < a.V) {
                    uint2 ob;
                    ob.x = pack_bf16x2(act_f(fmaf(bf_lo(yb[it].x), sc4[0], sh4[0]), a.nf.act), act_f(fmaf(bf_hi(yb[it].x), sc4[1], sh4[1]), a.nf.act));
                    ob.y = pack_bf16x2(act_f(fmaf(bf_lo(yb[it].y), sc4[2], sh4[2]), a.nf.act), act_f(fmaf(bf_hi(yb[it].y), sc4[3], sh4[3]), a.nf.act));
                    *(uint2*)(ao + ((size_t)m * C + c) * 2) = ob;
                }
            }
            __syncthreads();                                 // s_par / dsum are reused by the next row tile
        } else {
            // DEEP_BWD_NORM: the destination is dL/d(view) of a norm layer's tensor, complete with this launch
            char* gbuf = (char*)a.c.out[0];
            const char* u = (const char*)a.nb.u;
            const int oacc = a.c.out_acc[0];
            float mean4[4], rstd4[4], sc4[4], sh4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                mean4[r] = a.nb.stat[c + r]; rstd4[r] = a.nb.stat[C + c + r]; sc4[r] = a.nb.stat[2 * C + c + r]; sh4[r] = a.nb.stat[3 * C + c + r];
            }
            // the old gradient (if this launch accumulates) and the raw tensor of this thread's rows: requested together
            uint2 old[MAXIT], ub[MAXIT];
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
                const int m = it * 64 + ml;
                const bool ok = it < iters && m < a.V;
                old[it] = (ok && oacc) ? *(const uint2*)(gbuf + ((size_t)m * C + c) * 2) : make_uint2(0u, 0u);
                ub[it] = ok ? *(const uint2*)(u + ((size_t)m * C + c) * 2) : make_uint2(0u, 0u);
            }
            // dL/d(view) = sum (+ old), rounded as the separate passes would have stored it; sums of dv and dv * xhat
            uint2 gb[MAXIT];
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
                const int m = it * 64 + ml;
                gb[it] = make_uint2(0u, 0u);
                if (it < iters && m < a.V) {
                    gb[it].x = pack_bf16x2(sum[it][0] + bf_lo(old[it].x), sum[it][1] + bf_hi(old[it].x));
                    gb[it].y = pack_bf16x2(sum[it][2] + bf_lo(old[it].y), sum[it][3] + bf_hi(old[it].y));
                    const float gg[4] = {bf_lo(gb[it].x), bf_hi(gb[it].x), bf_lo(gb[it].y), bf_hi(gb[it].y)};
                    const float uu[4] = {bf_lo(ub[it].x), bf_hi(ub[it].x), bf_lo(ub[it].y), bf_hi(ub[it].y)};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float dv = gg[r] * act_d(fmaf(uu[r], sc4[r], sh4[r]), a.nb.act);
                        sa[r] += dv;
                        sb[r] = fmaf(dv, (uu[r] - mean4[r]) * rstd4[r], sb[r]);
                    }
                }
            }
            deep_reduce16(sa, sb, dsum);
            if (tid < 16) {                                  // k_norm_bwd_finalize's arithmetic
                const int cc = nt * 16 + tid;
                const float c0 = a.nb.gamma[cc] * a.nb.stat[C + cc], m1 = (float)(dsum[tid] / (double)a.V), m2 = (float)(dsum[16 + tid] / (double)a.V);
                a.nb.coef[cc] = c0; a.nb.coef[C + cc] = m1; a.nb.coef[2 * C + cc] = m2;
                a.nb.dgamma[cc] += (float)dsum[16 + tid];
                a.nb.dbeta[cc] += (float)dsum[tid];
                s_par[tid] = c0; s_par[16 + tid] = m1; s_par[32 + tid] = m2;
            }
            __syncthreads();
            // dL/d(raw) = A * dv + B * u + D (k_norm_bwd_finalize_apply8's form), written over the gradient
            float A4[4], B4[4], D4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float c0 = s_par[rq * 4 + r], m1 = s_par[16 + rq * 4 + r], m2 = s_par[32 + rq * 4 + r];
                A4[r] = c0;
                B4[r] = -c0 * rstd4[r] * m2;
                D4[r] = -c0 * (m1 - mean4[r] * rstd4[r] * m2);
            }
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
                const int m = it * 64 + ml;
                if (it < iters && m < a.V) {
                    const float gg[4] = {bf_lo(gb[it].x), bf_hi(gb[it].x), bf_lo(gb[it].y), bf_hi(gb[it].y)};
                    const float uu[4] = {bf_lo(ub[it].x), bf_hi(ub[it].x), bf_lo(ub[it].y), bf_hi(ub[it].y)};
                    float rr[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) rr[r] = fmaf(A4[r] * gg[r], act_d(fmaf(uu[r], sc4[r], sh4[r]), a.nb.act), fmaf(B4[r], uu[r], D4[r]));
                    uint2 ob;
                    ob.x = pack_bf16x2(rr[0], rr[1]); ob.y = pack_bf16x2(rr[2], rr[3]);
                    *(uint2*)(gbuf + ((size_t)m * C + c) * 2) = ob;
                }
            }
            __syncthreads();
        }
    }
}



template <int S, int KD, int PAD, bool SC, int EPI, int NTW>
__global__ void __launch_bounds__(256) k_deep_conv(DeepArgs a) {
    constexpr int T = KD * KD * KD, PF = 4;
    static_assert(EPI == DEEP_PLAIN || !SC, "a norm epilogue owns whole channels: no scatter kinds");
    __shared__ __attribute__((aligned(16))) float red[4 * 4 * NTW * 64 * 4];
    __shared__ double dsum[32 + 128];
    __shared__ float s_par[64];
    __shared__ int s_last;
    const ConvGeom& g = a.c.g;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, gq = lane >> 4;
    const int NTT = g.Cout / 16, NG = NTT / NTW, NGK = NG * a.ksplit;
    // blocks that read the same filter slice (same row tiles and K range, different voxel group) are NGK apart in the grid: with NGK a
    // multiple of 8 (16+ row tiles or a K split of 8+: every shape of the default architecture) they share an XCD and its L2
    const int ngk = blockIdx.x % NGK, mg = blockIdx.x / NGK, ng = ngk % NG, ksp = ngk / NG;
    const int KS = T * a.nchunk;
    const int k0 = ksp * a.kper, k1 = k0 + a.kper < KS ? k0 + a.kper : KS;
    const int per = (k1 - k0 + 3) >> 2;
    const int kw0 = k0 + wave * per, kw1 = kw0 + per < k1 ? kw0 + per : k1;

    int oz[4], oy[4], ox[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = mg * 64 + i * 16 + j;
        ox[i] = m % g.Wo;
        const int r = m / g.Wo;
        oy[i] = r % g.Ho;
        oz[i] = m < a.V ? r / g.Ho : -(1 << 20);       // a row past the grid reads zeros (every tap lands outside the volume)
    }
    const bf16x8* wp = (const bf16x8*)a.c.w + (size_t)(ng * NTW) * 64 + lane;
    const int C0 = a.c.src[0].C;
    f32x4 acc[4][NTW];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int n = 0; n < NTW; ++n) acc[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 xb[PF][4], wb[PF][NTW];
    // The K loop: PF k-steps of operands in flight in registers.  Every load is UNCONDITIONAL (a voxel outside the volume reads the
    // zero page, a k-step past the wave's range repeats the last one): with a branch around a load the compiler can no longer count the
    // loads in flight and waits for all of them (vmcnt(0)) before the first MFMA of every round -- one exposed memory latency per PF
    // k-steps (measured: 1.4 us per k-step at the 8^3 levels).  Straight-line rounds get counted waits (vmcnt((PF - 1) * (4 + NTW))).
    const int nst = kw1 > kw0 ? kw1 - kw0 : 0;
    int pt = nst ? kw0 / a.nchunk : 0, pq = nst ? kw0 - pt * a.nchunk : 0, rem = nst;   // the next k-step to request: (tap, chunk)

    // the input voxel of each of the lane's four rows for the tap being requested (-1: outside the volume), recomputed only when the tap
    // changes -- k-steps are tap-major, a wave's range covers one or two taps.  (With the voxel arithmetic in every k-step the kernel was
    // VALU-bound: three quarter-rate 32-bit multiplies per row and k-step, ~400 cycles per k-step against 64 of MFMA.)
    int vox[4] = {-1, -1, -1, -1};
    bool newtap = true;
#define DEEP_ISSUE(st)                                                                                                              \
    {                                                                                                                               \
        if (newtap) {                                                                                                               \
            const int kz = pt / (KD * KD), ky = (pt / KD) % KD, kx = pt % KD;                                                       \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                        \
                const int iz = S * oz[i] + kz - PAD, iy = S * oy[i] + ky - PAD, ix = S * ox[i] + kx - PAD;                          \
                const bool ok = (unsigned)iz < (unsigned)g.D && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;       \
                vox[i] = ok ? (int)__umul24(__umul24(iz, g.H) + iy, g.W) + ix : -1;                                                 \
            }                                                                                                                       \
            newtap = false;                                                                                                         \
        }                                                                                                                           \
        const int cch = pq * 32;                                                                                                    \
        const bool second = a.c.nsrc > 1 && cch >= C0;                                                                              \
        const char* sbase = (const char*)(second ? a.c.src[1].ptr : a.c.src[0].ptr) + ((second ? cch - C0 : cch) + gq * 8) * 2;     \
        const unsigned vstride = (unsigned)(second ? a.c.src[1].C : C0) * 2u;                                                       \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                            \
            const char* xp = vox[i] >= 0 ? sbase + __umul24((unsigned)vox[i], vstride) : (const char*)g_deep_zero;                  \
            xb[st][i] = *(const bf16x8*)xp;                                                                                         \
        }                                                                                                                           \
        _Pragma("unroll") for (int n = 0; n < NTW; ++n) wb[st][n] = wp[((size_t)(pq * T + pt) * NTT + n) * 64];                    \
        if (rem > 1) { --rem; if (++pq == a.nchunk) { pq = 0; ++pt; newtap = true; } }                                              \
    }
#define DEEP_MFMA(st)                                                                                                               \
    _Pragma("unroll") for (int n = 0; n < NTW; ++n)                                                                                \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                              \
            acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[st][n], xb[st][i], acc[i][n], 0, 0, 0);

#pragma unroll
    for (int st = 0; st < PF; ++st) DEEP_ISSUE(st)
    int kk = 0;
#pragma unroll 1
    for (; kk + PF < nst; kk += PF) {        // full rounds: every stage is a k-step of the range and is refilled
#pragma unroll
        for (int st = 0; st < PF; ++st) {
            DEEP_MFMA(st)
            DEEP_ISSUE(st)
        }
    }
#pragma unroll
    for (int st = 0; st < PF; ++st)           // the last round (1..PF k-steps): nothing left to request
        if (kk + st < nst) { DEEP_MFMA(st) }
#undef DEEP_MFMA
#undef DEEP_ISSUE

    // ---- the four waves' K quarters; wave w finishes m-tile w ----
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int n = 0; n < NTW; ++n) *(f32x4*)(red + (((wave * 4 + i) * NTW + n) * 64 + lane) * 4) = acc[i][n];
    __syncthreads();
    f32x4 o[NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
        o[n] = *(const f32x4*)(red + (((0 * 4 + wave) * NTW + n) * 64 + lane) * 4);
#pragma unroll
        for (int v = 1; v < 4; ++v) {
            const f32x4 t = *(const f32x4*)(red + (((v * 4 + wave) * NTW + n) * 64 + lane) * 4);
            o[n][0] += t[0]; o[n][1] += t[1]; o[n][2] += t[2]; o[n][3] += t[3];
        }
    }
    const int mw = mg * 64 + wave * 16 + j;
    if (EPI == DEEP_PLAIN && a.ksplit == 1) {
#pragma unroll
        for (int n = 0; n < NTW; ++n) deep_store<SC>(a, mw, (ng * NTW + n) * 16 + gq * 4, o[n]);
        return;
    }
    // ---- publish the partial tile, take a ticket ----
#pragma unroll
    for (int n = 0; n < NTW; ++n)
        deep_publish(a.part + (((size_t)(ksp * NTT + ng * NTW + n) * a.Vpad + mw) * 16 + gq * 4), o[n]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // this wave's stores have completed (no cache maintenance at this scope)
    __syncthreads();
    if (tid == 0) {
        const int cidx = EPI == DEEP_PLAIN ? mg * NG + ng : ng, target = EPI == DEEP_PLAIN ? a.ksplit : a.MG * a.ksplit;
        const int old = __hip_atomic_fetch_add(a.cnt + cidx, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = old == target - 1;
        if (s_last) __hip_atomic_store(a.cnt + cidx, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
    deep_finish<SC, EPI, NTW, true>(a, mg, ng, dsum, s_par);
}

// ---- launch plumbing ----
// UNET_NO_DEEP_KERNELS=1 (read once per process): the halo-tile kernels and the separate norm launches at the deep levels too (also
// implied by UNET_NO_SLIDING_WINDOW); tests/test_gpu_parity.py compares the two paths in the network
static bool deep_off() {
    static const bool off = getenv("UNET_NO_DEEP_KERNELS") != nullptr;
    return off || sliding_window_off();
}
// the K split: enough blocks to stream the filter from every CU, the partial tiles inside the scratch
static int deep_ksplit(int KS, int MG, int NG, int NTT, int Vpad, const DeepScratch& sc) {
    // a wave's share of K in at most two rounds of its 4-deep prefetch (each round is one exposed memory latency), with up to 1024
    // blocks and 4 MB of partial tiles; small grids go on splitting down to one round per wave
    const size_t cap = sc.part_bytes < ((size_t)4 << 20) ? sc.part_bytes : ((size_t)4 << 20);
    auto fits = [&](int k) { return k <= 32 && (int64_t)MG * NG * k <= 1024 && (size_t)k * NTT * Vpad * 16 * 4 <= cap; };
    int ks = 1;
    while (KS / (4 * ks) > 7 && fits(2 * ks)) ks *= 2;
    while ((int64_t)MG * NG * ks < 256 && KS / (4 * ks) > 3 && fits(2 * ks)) ks *= 2;
    return ks;
}
template <int S, int KD, int PAD, bool SC>
static bool deep_launch(MfmaConvArgs c, int epi, const DeepNormFwd* nf, const DeepNormBwd* nb, const DeepScratch& sc, hipStream_t s) {
    DeepArgs a;
    a.c = c;
    const ConvGeom& g = c.g;
    a.V = g.Do * g.Ho * g.Wo; a.Vpad = (a.V + 63) / 64 * 64; a.MG = a.Vpad / 64;
    a.nchunk = g.Cin / 32;
    const int T = KD * KD * KD, KS = T * a.nchunk, NTT = g.Cout / 16;
    const int ntw = (NTT % 2 == 0 && (int64_t)a.MG * (NTT / 2) >= 128) ? 2 : 1;   // two row tiles per block only when the grid stays wide
    const int NG = NTT / ntw;
    a.ksplit = deep_ksplit(KS, a.MG, NG, NTT, a.Vpad, sc);
    a.kper = (KS + a.ksplit - 1) / a.ksplit;
    a.ksplit = (KS + a.kper - 1) / a.kper;
    if ((size_t)a.ksplit * NTT * a.Vpad * 16 * 4 > sc.part_bytes) return false;
    const int ncnt = epi == DEEP_PLAIN ? a.MG * NG : NG;
    if (ncnt > sc.ncnt) return false;
    a.part = sc.part; a.cnt = sc.cnt;
    if (nf) a.nf = *nf;
    if (nb) a.nb = *nb;
    const unsigned grid = (unsigned)(a.MG * NG * a.ksplit);
#define DEEP_GO(EPI)                                                                                  \
    do {                                                                                              \
        if (ntw == 2) k_deep_conv<S, KD, PAD, SC, EPI, 2><<<grid, 256, 0, s>>>(a);                    \
        else k_deep_conv<S, KD, PAD, SC, EPI, 1><<<grid, 256, 0, s>>>(a);                             \
    } while (0)
    if constexpr (SC) { DEEP_GO(DEEP_PLAIN); }
    else {
        if (epi == DEEP_FWD_NORM) DEEP_GO(DEEP_FWD_NORM);
        else if (epi == DEEP_BWD_NORM) DEEP_GO(DEEP_BWD_NORM);
        else DEEP_GO(DEEP_PLAIN);
    }
#undef DEEP_GO
    return true;
}
static bool deep_src_ok(const SrcDesc* src, int nsrc) {
    for (int k = 0; k < nsrc; ++k)
        if (src[k].C % 32 || src[k].scale || src[k].act) return false;
    return true;
}
static MfmaConvArgs deep_base() {
    MfmaConvArgs a;
    a.nsrc = 1; a.w = nullptr; a.bias = nullptr;
    a.out[0] = a.out[1] = nullptr; a.outC[0] = a.outC[1] = 0; a.out_acc[0] = a.out_acc[1] = 0; a.nout = 1;
    a.stats = nullptr; a.tiles_x = a.tiles_y = a.tiles_z = 0; a.sc_C = 0; a.oD = a.oH = a.oW = 0;
    return a;
}
// which output grids these kernels take: the levels where a 64-voxel tile x 16 rows grid cannot fill the chip
// Measured per layer of the default architecture at 128^3 (profiles/r20c_step_launch_sequence.txt against r18_step_launch_sequence.txt):
// at 4^3 a 3x3x3 conv + norm takes 11.5 us here against 20-30 us in two launches; at 8^3 it takes 25-41 us against 20-29 -- with
// 512..1024 blocks the ticket path (stores acknowledged by the memory side, the atomic's round trip, the last arriver's four rounds of
// loads over 512 voxels, a second wave of blocks) costs more than the split saves; and as two launches (partial tiles with plain stores
// from 1024 lean blocks, then a finish launch of 16 blocks: profiles/r20d_two_launch_8cube_not_kept_launch_sequence.txt) 17 + 10 us.
// So the 27-tap kinds come here at 64 voxels or fewer;
// the short contractions (conv_trans forward and dgrad, the stride-2 dgrad: 1 or 8 taps) win up to DEEP_MAX_VOXELS.
constexpr int64_t DEEP_K3_MAX_VOXELS = 64;
bool deep_conv_applies(int dtype, int64_t out_voxels, int cin, int cout) {
    return !deep_off() && dtype == 1 && out_voxels <= DEEP_K3_MAX_VOXELS && cin % 32 == 0 && cout % 16 == 0;
}
static bool deep_short_applies(int64_t grid_voxels, int cin, int cout) {
    return !deep_off() && grid_voxels <= DEEP_MAX_VOXELS && cin % 32 == 0 && cout % 16 == 0;
}

bool launch_deep_conv_fwd(const ConvGeom& g, const SrcDesc* src, int nsrc, const void* w_mfma, const float* bias, void* out,
                          const DeepNormFwd* nf, const DeepScratch& sc, hipStream_t s) {
    if (!sc.part || !deep_conv_applies(1, (int64_t)g.Do * g.Ho * g.Wo, g.Cin, g.Cout) || g.ks != 3 || !deep_src_ok(src, nsrc)) return false;
    if (g.stride == 2 && g.Wo > 8) return false;                          // the stride-2 forward pack has 32-channel chunks at Wo <= 8 only
    MfmaConvArgs a = deep_base();
    a.g = g; a.nsrc = nsrc; a.src[0] = src[0]; if (nsrc > 1) a.src[1] = src[1];
    a.w = w_mfma; a.bias = bias;
    a.out[0] = out; a.outC[0] = g.Cout;
    a.oD = g.Do; a.oH = g.Ho; a.oW = g.Wo;
    const int epi = nf ? DEEP_FWD_NORM : DEEP_PLAIN;
    return g.stride == 1 ? deep_launch<1, 3, 1, false>(a, epi, nf, nullptr, sc, s) : deep_launch<2, 3, 1, false>(a, epi, nf, nullptr, sc, s);
}
bool launch_deep_convt_fwd(const ConvGeom& g, const SrcDesc* src, int nsrc, const void* w_mfma, const float* bias, void* out,
                           const DeepScratch& sc, hipStream_t s) {
    if (!sc.part || !deep_short_applies((int64_t)g.D * g.H * g.W, g.Cin, g.Cout) || !deep_src_ok(src, nsrc)) return false;
    MfmaConvArgs a = deep_base();
    a.g = g; a.g.Cout = 8 * g.Cout; a.g.Do = g.D; a.g.Ho = g.H; a.g.Wo = g.W; a.g.ks = 1; a.g.stride = 1;
    a.nsrc = nsrc; a.src[0] = src[0]; if (nsrc > 1) a.src[1] = src[1];
    a.w = w_mfma; a.bias = bias;
    a.out[0] = out; a.outC[0] = g.Cout;
    a.sc_C = g.Cout; a.oD = g.Do; a.oH = g.Ho; a.oW = g.Wo;
    return deep_launch<1, 1, 0, true>(a, DEEP_PLAIN, nullptr, nullptr, sc, s);
}
// returns 0: shape not served; 1: gradient written / accumulated; 2: ... and the destination's norm backward is done (nb given, one destination)
int launch_deep_conv_dgrad(const ConvGeom& g, const void* dy, const void* w_mfma_dgrad, const DstGrad* dst, int ndst, const DeepNormBwd* nb,
                           const DeepScratch& sc, hipStream_t s) {
    if (!sc.part || deep_off() || g.ks != 3 || g.Cout % 32 || g.Cin % 16) return 0;
    MfmaConvArgs a = deep_base();
    a.src[0].ptr = dy; a.src[0].C = g.Cout;
    a.w = w_mfma_dgrad;
    for (int k = 0; k < 2; ++k) {
        a.out[k] = k < ndst ? dst[k].ptr : nullptr;
        a.outC[k] = k < ndst ? dst[k].C : 0;
        a.out_acc[k] = k < ndst ? dst[k].accumulate : 0;
    }
    a.nout = ndst;
    a.g.Cin = g.Cout; a.g.D = g.Do; a.g.H = g.Ho; a.g.W = g.Wo; a.g.ks = 3; a.g.stride = 1;
    a.oD = g.D; a.oH = g.H; a.oW = g.W;
    if (g.stride == 1) {
        if ((int64_t)g.D * g.H * g.W > DEEP_K3_MAX_VOXELS) return 0;
        a.g.Cout = g.Cin; a.g.Do = g.D; a.g.Ho = g.H; a.g.Wo = g.W;
        const bool fuse = nb && ndst == 1 && dst[0].ptr && dst[0].C == g.Cin;
        return deep_launch<1, 3, 1, false>(a, fuse ? DEEP_BWD_NORM : DEEP_PLAIN, nullptr, fuse ? nb : nullptr, sc, s) ? (fuse ? 2 : 1) : 0;
    }
    // stride 2: 8 taps over dL/dy on the coarse grid, rows = 8 output parities x Cin, scattered to 2 * m + parity
    a.g.Cout = 8 * g.Cin; a.sc_C = g.Cin;
    a.g.Do = (g.D + 1) / 2; a.g.Ho = (g.H + 1) / 2; a.g.Wo = (g.W + 1) / 2;
    if ((int64_t)a.g.Do * a.g.Ho * a.g.Wo > DEEP_MAX_VOXELS) return 0;
    return deep_launch<1, 2, 0, true>(a, DEEP_PLAIN, nullptr, nullptr, sc, s) ? 1 : 0;
}
bool launch_deep_convt_dgrad(const ConvGeom& g, const void* dy, const void* w_mfma_dgrad, const DstGrad* dst, int ndst, const DeepNormBwd* nb,
                             const DeepScratch& sc, int* norm_done, hipStream_t s) {
    if (norm_done) *norm_done = 0;
    if (!sc.part || deep_off() || g.Cout % 32 || g.Cin % 16 || g.W > 8 || (int64_t)g.D * g.H * g.W > DEEP_MAX_VOXELS) return false;
    MfmaConvArgs a = deep_base();
    a.src[0].ptr = dy; a.src[0].C = g.Cout;
    a.w = w_mfma_dgrad;
    for (int k = 0; k < 2; ++k) {
        a.out[k] = k < ndst ? dst[k].ptr : nullptr;
        a.outC[k] = k < ndst ? dst[k].C : 0;
        a.out_acc[k] = k < ndst ? dst[k].accumulate : 0;
    }
    a.nout = ndst;
    a.g.Cin = g.Cout; a.g.Cout = g.Cin; a.g.D = g.Do; a.g.H = g.Ho; a.g.W = g.Wo; a.g.Do = g.D; a.g.Ho = g.H; a.g.Wo = g.W;
    a.g.ks = 2; a.g.stride = 2;
    a.oD = g.D; a.oH = g.H; a.oW = g.W;
    const bool fuse = nb && ndst == 1 && dst[0].ptr && dst[0].C == g.Cin;
    if (!deep_launch<2, 2, 0, false>(a, fuse ? DEEP_BWD_NORM : DEEP_PLAIN, nullptr, fuse ? nb : nullptr, sc, s)) return false;
    if (norm_done) *norm_done = fuse ? 1 : 0;
    return true;
}

}  // namespace unet
