// Sliding-window weight gradients of the layers that cross a resolution boundary, coarse grid >= 24 wide (gfx950, bf16, fp32 accumulate):
//
//   Conv3d(k3, s2, p1):          dW[co][ci][t] = sum_m  x[2m + t - 1][ci] * dy[m][co]       fine = x,  coarse = dy      (unet.cpp:59-72, train.cpp:706)
//   ConvTranspose3d(k2, s2):     dW[ci][co][t] = sum_m  x[m][ci] * dy[2m + t][co]           fine = dy, coarse = x       (unet.cpp:46-57)
//
// Both are T implicit GEMMs  D_t[ca][cb] += A_t[ca][k] * B[k][cb]  with k = COARSE voxel, A = the fine tensor read at 2k + t - pad
// (ca = its channels), B = the coarse tensor -- the conventions of k_mfma_wgrad (kernels_mfma_wgrad.hip), whose halo-tile form these
// replace (measured there at 128^3 <-> 64^3: 2.06 TB/s, 1.52x the algorithmic traffic, 68 % of the wave cycles parked, a third of the
// LDS cycles bank conflicts of the stride-2 transposing reads).  Here a block owns a (BYC x 32) coarse footprint and walks z like
// k_mfma_wgrad_z; every step brings ONE coarse plane tile and TWO fine planes by LDS-DMA (global_load_lds_dwordx4), one step ahead
// into rings of 2 / 5 (k2: 4) slots -- the wait is a plain vmcnt(0): nothing else of the wave is in flight, no counting.  A fine
// plane's LDS image is space-to-depth in x (a row = its odd-x voxels, then its even-x voxels; k2: even, odd), so the 32 coarse voxels of
// a K-step read 32 CONSECUTIVE fine voxels for every tap and both operands use the same conflict-free ds_read_b64_tr_b16 pattern.
// The kernel is HBM-bound by an order of magnitude (7 GFLOP for 84 MB), so an A fragment is simply read per tap.
// Slab per block in the gradient's layout [cb][ca][t] (+ bias partial row), summed in a fixed order by the reduce kernels.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "mfma_util.h"

namespace unet {

__device__ __attribute__((aligned(16))) unsigned g_s2w_zero[4] = {0u, 0u, 0u, 0u};

typedef __attribute__((ext_vector_type(4))) short ws16x4;
typedef __attribute__((address_space(3))) ws16x4 wlds_s16x4;
__device__ __forceinline__ bf16x8 wtr_read2(const char* p0, const char* p1) {
    ws16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wlds_s16x4*)p0);
    ws16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wlds_s16x4*)p1);
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

struct S2WgradArgs {
    const void* fine;     // [fD][fH][fW][Ca] bf16; a channel concat {fine (C0 channels), fine2} when fine2 != nullptr
    const void* fine2;
    int C0;
    const void* coarse;   // [cD][cH][cW][Cb] bf16
    int Ca, Cb;
    int fD, fH, fW, cD, cH, cW;
    float* slab;          // [gridDim.x][Cb][Ca][T]
    float* bias_slab;     // [gridDim.x][Cb] (sums of the coarse tensor) or [gridDim.x][Ca] (bias_from_a: sums of the fine tensor), or nullptr
    int bias_from_a;
    int cols_x, cols_y, nseg, zlen;
};

#define S2W_DMA(src, dst)                                                                                                    \
    do {                                                                                                                     \
        unsigned keep_;                                                                                                      \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_) : "v"(src), "s"(__builtin_amdgcn_readfirstlane((int)(dst))) : "memory");                      \
    } while (0)

template <int KS, int PA, int PB, int WK>
__global__ void __launch_bounds__(256, 2) k_s2_wgrad(S2WgradArgs a) {
    constexpr int T = KS * KS * KS, PAD = KS == 3 ? 1 : 0, P = PA * PB;
    static_assert(P * WK == 4, "four waves");
    constexpr int R = WK == 1 ? 2 : 1, BYC = WK * R;                   // K-steps (coarse rows of 32) per wave and plane; coarse rows per block
    constexpr int HYF = 2 * BYC + PAD, NODD = KS == 3 ? 33 : 32, HXF = NODD + 32;
    constexpr int FVS = PA * 32, CVS = PB * 32;                        // LDS voxel strides: [voxel][tile][16 ch]
    constexpr int FSLOT = HYF * HXF * FVS, CSLOT = BYC * 32 * CVS;
    constexpr int NF = KS == 3 ? 5 : 4, NC = 2;
    constexpr int FUNITS = HYF * HXF * PA * 2, CUNITS = BYC * 32 * PB * 2;
    constexpr int ITF = (FUNITS + 255) / 256, ITC = (CUNITS + 255) / 256;
    constexpr int GA = PA * 2, GB = PB * 2;                            // a thread always stages the same (tile, 8-channel half): tid % GA, tid % GB
    constexpr int LDS_BYTES = NF * FSLOT + NC * CSLOT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const fring = smem;
    char* const cring = smem + NF * FSLOT;

    const int tid = threadIdx.x, lane = tid & 63, il = lane & 15, gq = lane >> 4, q4 = il >> 2, p4 = il & 3;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), pw = wave % P, kw = wave / P, ia = pw / PB, ib = pw % PB;
    const int CBG = (a.Cb / 16) / PB;
    const int caB = ((int)blockIdx.y / CBG) * PA, cbB = ((int)blockIdx.y % CBG) * PB;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    const int item = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int ncols = a.cols_x * a.cols_y, seg = item / ncols, col = item % ncols;
    const int x0 = (col % a.cols_x) * 32, y0 = (col / a.cols_x) * BYC;
    const int zs = seg * a.zlen, ze = zs + a.zlen < a.cD ? zs + a.zlen : a.cD, len = ze - zs;

    f32x4 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum[e] = 0.f;
    const bool do_bias_b = a.bias_slab != nullptr && !a.bias_from_a && caB == 0;
    const bool do_bias_a = a.bias_slab != nullptr && a.bias_from_a && cbB == 0;

    // ---- staging units (fixed per thread): byte offset inside a plane of the source, or ~0 outside the volume ----
    unsigned foff[ITF], coff[ITC];
    unsigned fpiece[ITF], cpiece[ITC];
    // this thread's 8 fine channels come from one source of the concat: its pointer, channel count and channel offset
    const int fch = caB * 16 + (tid % GA) * 8;
    const bool f2 = a.fine2 != nullptr && fch >= a.C0;
    const char* const fptr = (const char*)(f2 ? a.fine2 : a.fine);
    const int fC = a.fine2 ? (f2 ? a.Ca - a.C0 : a.C0) : a.Ca, fc0 = f2 ? fch - a.C0 : fch;
#pragma unroll
    for (int it = 0; it < ITF; ++it) {
        const int u = it * 256 + tid, hv = u / GA, hy = hv / HXF, sl = hv % HXF;
        const int t = sl < NODD ? 2 * sl : 2 * (sl - NODD) + 1;
        const int gy = 2 * y0 - PAD + hy, gx = 2 * x0 - PAD + t;
        const bool ok = u < FUNITS && (unsigned)gy < (unsigned)a.fH && (unsigned)gx < (unsigned)a.fW;
        foff[it] = ok ? (unsigned)(((size_t)gy * a.fW + gx) * fC + fc0) * 2u : 0xffffffffu;
        if (u >= FUNITS) foff[it] = 0xfffffffeu;      // no unit at all: the lane stays out of the DMA
        fpiece[it] = (unsigned)__builtin_amdgcn_readfirstlane((it * 256 + wave * 64) * 16);
    }
#pragma unroll
    for (int it = 0; it < ITC; ++it) {
        const int u = it * 256 + tid, tv = u / GB, th = u % GB, ty = tv / 32, tx = tv % 32;
        const int gy = y0 + ty, gx = x0 + tx;
        const bool ok = u < CUNITS && gy < a.cH && gx < a.cW;
        coff[it] = ok ? (unsigned)(((size_t)gy * a.cW + gx) * a.Cb + (cbB * 16 + th * 8)) * 2u : 0xffffffffu;
        if (u >= CUNITS) coff[it] = 0xfffffffeu;
        cpiece[it] = (unsigned)__builtin_amdgcn_readfirstlane((it * 256 + wave * 64) * 16);
    }
    const size_t fplane = (size_t)a.fH * a.fW * fC * 2, cplane = (size_t)a.cH * a.cW * a.Cb * 2;
    // fine plane index f (relative to fine z = 2 zs - PAD) -> ring slot f % NF; coarse plane n -> slot n % NC
    auto dma_fine = [&](int f) {
        const int fz = 2 * zs - PAD + f;
        const bool zin = (unsigned)fz < (unsigned)a.fD;
        const unsigned slot = (unsigned)(f % NF);
#pragma unroll
        for (int it = 0; it < ITF; ++it) {
            const char* src = (zin && foff[it] < 0xfffffffeu) ? fptr + (size_t)fz * fplane + foff[it] : (const char*)g_s2w_zero;
            if (foff[it] != 0xfffffffeu) S2W_DMA(src, lds0 + slot * FSLOT + fpiece[it]);
        }
    };
    auto dma_coarse = [&](int n) {
        const int cz = zs + n;
        const bool zin = cz < ze;
        const unsigned slot = (unsigned)(n % NC);
#pragma unroll
        for (int it = 0; it < ITC; ++it) {
            const char* src = (zin && coff[it] < 0xfffffffeu) ? (const char*)a.coarse + (size_t)cz * cplane + coff[it] : (const char*)g_s2w_zero;
            if (coff[it] != 0xfffffffeu) S2W_DMA(src, lds0 + NF * FSLOT + slot * CSLOT + cpiece[it]);
        }
    };
    // per-channel sums of the units this thread staged (read back from LDS once they have landed)
    auto sum_units = [&](const char* slotp, int iters, int units) {
        for (int it = 0; it < iters; ++it) {
            const int u = it * 256 + tid;
            if (u < units) {
                const uint4 v = *(const uint4*)(slotp + (size_t)u * 16);
                bsum[0] += bf_lo(v.x); bsum[1] += bf_hi(v.x); bsum[2] += bf_lo(v.y); bsum[3] += bf_hi(v.y);
                bsum[4] += bf_lo(v.z); bsum[5] += bf_hi(v.z); bsum[6] += bf_lo(v.w); bsum[7] += bf_hi(v.w);
            }
        }
    };

    // ---- fragment addresses: lane group gq, read r fetch voxel group G = gq + 4r of the K-step (4 consecutive voxels), lane 4q+p
    // supplies voxel q, channels 4p..4p+3 (the transposing read hands lane il channel il of 8 voxels) ----
    int aoff[2], boff[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int v = 4 * (gq + 4 * r) + q4;
        aoff[r] = v * FVS + ia * 32 + p4 * 8;
        boff[r] = v * CVS + ib * 32 + p4 * 8;
    }

    // prologue: everything step 0 reads
    if (len > 0) {
        if constexpr (KS == 3) dma_fine(0);
        dma_fine(KS == 3 ? 1 : 0);
        dma_fine(KS == 3 ? 2 : 1);
        dma_coarse(0);
    }
    for (int n = 0; n < len; ++n) {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");     // step n's planes have landed; every wave is done with step n - 1
        if (n + 1 < len) {                                                 // one step ahead, into the slots step n does not read
            dma_fine(2 * (n + 1) + (KS == 3 ? 1 : 0));
            dma_fine(2 * (n + 1) + (KS == 3 ? 2 : 1));
            dma_coarse(n + 1);
        }
        const char* cs = cring + (n % NC) * CSLOT;
        if (do_bias_b) sum_units(cs, ITC, CUNITS);
        if (do_bias_a) {       // k2: the two fine planes of this step are staged exactly once in the whole launch (no halo)
            sum_units(fring + ((2 * n) % NF) * FSLOT, ITF, FUNITS);
            sum_units(fring + ((2 * n + 1) % NF) * FSLOT, ITF, FUNITS);
        }
        const char* fs[KS];
#pragma unroll
        for (int kz = 0; kz < KS; ++kz) fs[kz] = fring + ((2 * n + kz) % NF) * FSLOT;
#pragma unroll
        for (int jr = 0; jr < R; ++jr) {
            const int cr = kw * R + jr;
            const bf16x8 Bf = wtr_read2(cs + cr * 32 * CVS + boff[0], cs + cr * 32 * CVS + boff[1]);
            constexpr int RD = 4;
            bf16x8 ring[RD];
            auto afrag = [&](int t) {
                const int kz = t / (KS * KS), ky = (t / KS) % KS, kx = t % KS;
                const int segb = (kx & 1) ? NODD + (kx >> 1) : (kx >> 1);
                const char* p = fs[kz] + ((2 * cr + ky) * HXF + segb) * FVS;
                return wtr_read2(p + aoff[0], p + aoff[1]);
            };
#pragma unroll
            for (int t = 0; t < RD && t < T; ++t) ring[t] = afrag(t);
#pragma unroll
            for (int t = 0; t < T; ++t) {
                __builtin_amdgcn_sched_barrier(0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[t % RD], Bf, acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (t + RD < T) ring[t % RD] = afrag(t + RD);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- sum the K-split waves of each pair through LDS (taps in chunks that fit), then the slab (as k_mfma_wgrad_z) ----
    constexpr int TCH = (P * T * 1024 <= LDS_BYTES) ? T : ((P * 9 * 1024 <= LDS_BYTES && T % 9 == 0) ? 9 : (T % 3 == 0 ? 3 : 1));
    static_assert(P * TCH * 1024 <= LDS_BYTES && T % TCH == 0, "reduction scratch");
    float* red = (float*)smem;   // [P][TCH][64][4]
    if constexpr (WK > 1) {
#pragma unroll 1
        for (int kk = 1; kk < WK; ++kk) {
#pragma unroll
            for (int c0 = 0; c0 < T; c0 += TCH) {
                if (kw == kk) {
#pragma unroll
                    for (int t = 0; t < TCH; ++t) *(f32x4*)(red + ((pw * TCH + t) * 64 + lane) * 4) = acc[c0 + t];
                }
                __syncthreads();
                if (kw == 0) {
#pragma unroll
                    for (int t = 0; t < TCH; ++t) {
                        const f32x4 o = *(const f32x4*)(red + ((pw * TCH + t) * 64 + lane) * 4);
                        acc[c0 + t][0] += o[0]; acc[c0 + t][1] += o[1]; acc[c0 + t][2] += o[2]; acc[c0 + t][3] += o[3];
                    }
                }
                __syncthreads();
            }
        }
    }
    {   // a lane owns cb = il and ca = gq*4 .. +3: the pair's tile is transposed through LDS and leaves as whole [cb] rows of 16 T floats
        constexpr int RP = 16 * T + 4;
        static_assert(16 * RP * 4 <= LDS_BYTES, "slab staging");
        float* stg = (float*)smem;
#pragma unroll 1
        for (int pr = 0; pr < P; ++pr) {
            __syncthreads();
            if (kw == 0 && pw == pr) {
#pragma unroll
                for (int t = 0; t < T; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) stg[il * RP + (gq * 4 + r) * T + t] = acc[t][r];
            }
            __syncthreads();
            float* base = a.slab + (size_t)blockIdx.x * T * a.Ca * a.Cb + ((size_t)(cbB + pr % PB) * 16 * a.Ca + (size_t)(caB + pr / PB) * 16) * T;
            for (int q = tid; q < 16 * 4 * T; q += 256) {
                const int row = q / (4 * T), c4 = q % (4 * T);
                *(f32x4*)(base + (size_t)row * a.Ca * T + c4 * 4) = *(const f32x4*)(stg + row * RP + c4 * 4);
            }
        }
    }
    if (do_bias_a || do_bias_b) {
        // threads with equal tid % G hold partial sums of the same 8 channels: shuffle tree inside each wave, then the 4 wave totals through LDS
        const int G = do_bias_a ? GA : GB;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = bsum[e];
            for (int m = G; m < 64; m <<= 1) v += __shfl_xor(v, m);
            bsum[e] = v;
        }
        __syncthreads();
        float* bred = (float*)smem;   // [4][G][8]
        if (lane < G) {
#pragma unroll
            for (int e = 0; e < 8; ++e) bred[(wave * G + lane) * 8 + e] = bsum[e];
        }
        __syncthreads();
        if (tid < G * 8) {
            const int u = tid / 8, e = tid % 8;
            float sacc = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) sacc += bred[(w * G + u) * 8 + e];
            const int C = do_bias_a ? a.Ca : a.Cb, tB = do_bias_a ? caB : cbB;
            a.bias_slab[(size_t)blockIdx.x * C + (tB + (u >> 1)) * 16 + (u & 1) * 8 + e] = sacc;
        }
    }
}

// ---- configuration / launch ----
struct S2WCfg { int pa, pb, wk, byc, cols_x, cols_y, nseg, zlen, gx, gy; };
// kind 1: conv stride 2 (Ca = Cin, Cb = Cout), kind 2: conv_trans (Ca = Cout, Cb = Cin); cD/cH/cW = the coarse grid
static bool s2w_cfg(int Ca, int Cb, int cD, int cH, int cW, S2WCfg& c) {
    if (sliding_window_off() || Ca % 16 || Cb % 16 || cW < 24 || cD < 4) return false;
    if ((size_t)(2 * cH + 1) * (2 * cW + 1) * Ca * 2 >= ((size_t)1 << 31) || (size_t)cH * cW * Cb * 2 >= ((size_t)1 << 31)) return false;   // 32-bit offsets inside a plane
    const int cat = Ca / 16, cbt = Cb / 16;
    if (cat % 2 == 0 && cbt % 2 == 0) { c.pa = 2; c.pb = 2; c.wk = 1; }
    else if (cbt % 2 == 0) { c.pa = 1; c.pb = 2; c.wk = 2; }
    else if (cat % 2 == 0) { c.pa = 2; c.pb = 1; c.wk = 2; }
    else return false;
    c.byc = 2;
    c.cols_x = (cW + 31) / 32; c.cols_y = (cH + c.byc - 1) / c.byc;
    c.gy = (cat / c.pa) * (cbt / c.pb);
    const int cols = c.cols_x * c.cols_y;
    int want = 512 / c.gy;       // (128 / 256 / 512 blocks measured within noise of each other on the step: 2.706-2.715 ms)
    if (want < 1) want = 1;
    int nseg = (want + cols - 1) / cols;
    if (nseg < 1) nseg = 1;
    int zlen = (cD + nseg - 1) / nseg;
    if (zlen < 4) zlen = 4;
    if (zlen > cD) zlen = cD;
    c.nseg = (cD + zlen - 1) / zlen; c.zlen = zlen;
    c.gx = cols * c.nseg;
    return true;
}
int s2_wgrad_splits(int Ca, int Cb, int cD, int cH, int cW) {
    S2WCfg c;
    return s2w_cfg(Ca, Cb, cD, cH, cW, c) ? c.gx : 0;
}
template <int KS, int PA, int PB, int WK>
static void launch_s2w_t(const S2WgradArgs& a, const S2WCfg& c, hipStream_t s, int polite) {
    constexpr int PAD = KS == 3 ? 1 : 0, R = WK == 1 ? 2 : 1, BYC = WK * R, HYF = 2 * BYC + PAD, HXF = (KS == 3 ? 33 : 32) + 32;
    constexpr int lds = (KS == 3 ? 5 : 4) * HYF * HXF * PA * 32 + 2 * BYC * 32 * PB * 32;
    static_assert(BYC == 2 && lds <= 160 * 1024, "footprint / LDS");
    static std::atomic<uint64_t> attr_done{0};
    const int want = lds > 83000 ? lds : polite_lds(lds, 1);
    set_max_lds_once(attr_done, (const void*)k_s2_wgrad<KS, PA, PB, WK>, want);
    k_s2_wgrad<KS, PA, PB, WK><<<dim3((unsigned)c.gx, (unsigned)c.gy), 256, polite_lds(lds, polite), s>>>(a);
}
// kernel only: slab [gx][Cb][Ca][T] (+ bias partial rows behind it) at `scratch`; returns the number of rows, 0 = shape not served
int launch_s2_wgrad(int ks, const void* fine, const void* fine2, int C0, int Ca, int fD, int fH, int fW, const void* coarse, int Cb, int cD, int cH,
                    int cW, bool want_bias, int bias_from_a, void* scratch, hipStream_t s, int polite) {
    S2WCfg c;
    if (!s2w_cfg(Ca, Cb, cD, cH, cW, c)) return 0;
    S2WgradArgs a;
    a.fine = fine; a.fine2 = fine2; a.C0 = C0; a.coarse = coarse; a.Ca = Ca; a.Cb = Cb; a.fD = fD; a.fH = fH; a.fW = fW; a.cD = cD; a.cH = cH; a.cW = cW;
    const int T = ks * ks * ks;
    a.slab = (float*)scratch;
    a.bias_slab = want_bias ? a.slab + (size_t)c.gx * T * Ca * Cb : nullptr;
    a.bias_from_a = bias_from_a;
    a.cols_x = c.cols_x; a.cols_y = c.cols_y; a.nseg = c.nseg; a.zlen = c.zlen;
    if (ks == 3) {
        if (c.pa == 2 && c.pb == 2) launch_s2w_t<3, 2, 2, 1>(a, c, s, polite);
        else if (c.pb == 2) launch_s2w_t<3, 1, 2, 2>(a, c, s, polite);
        else launch_s2w_t<3, 2, 1, 2>(a, c, s, polite);
    } else {
        if (c.pa == 2 && c.pb == 2) launch_s2w_t<2, 2, 2, 1>(a, c, s, polite);
        else if (c.pb == 2) launch_s2w_t<2, 1, 2, 2>(a, c, s, polite);
        else launch_s2w_t<2, 2, 1, 2>(a, c, s, polite);
    }
    return c.gx;
}

}  // namespace unet
