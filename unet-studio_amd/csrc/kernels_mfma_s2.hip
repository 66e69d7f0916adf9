// Stride-2 family at the levels where it moves real bytes (coarse grid >= 16^3): sliding-window / LDS-DMA kernels in the style of
// k_mfma_conv_z16 (kernels_mfma_conv_z16.hip) for the four contractions that cross a resolution boundary
//
//   k_s2_scatter<TR = 0>   Conv3d(k3, s2, p1) input gradient      dx[2m+p] = sum_{d} W_{p,d} dy[m+d]         coarse -> fine   (unet.cpp:59-72, train.cpp:706)
//   k_s2_scatter<TR = 1>   ConvTranspose3d(k2, s2) forward        y[2m+p]  = W_p x[m] + b                     coarse -> fine   (unet.cpp:46-57)
//   k_s2_gather<KS = 3>    Conv3d(k3, s2, p1) forward             y[m]     = sum_k W_k x[2m+k-1] + b          fine -> coarse
//   k_s2_gather<KS = 2>    ConvTranspose3d(k2, s2) input gradient dx[m]    = sum_p W_p^T dy[2m+p]             fine -> coarse
//
// They replace the halo-tile kernel k_mfma_conv_p for these shapes (measured there, profiles/r12_*: 0.26 of the HBM roofline, 1.5x
// the algorithmic traffic, a third of the LDS cycles bank conflicts of the stride-2 gathers).  All four are HBM-bound by an order of
// magnitude (7 GFLOP against 84-218 MB at 128^3 <-> 64^3), so the design is about bytes in flight and whole-line accesses, not about
// MFMA issue: a block owns a small (y, x) footprint and walks z; every plane travels HBM -> LDS by LDS-DMA (global_load_lds_dwordx4,
// PF planes ahead into a ring), is awaited with ONE hand-counted s_waitcnt vmcnt((P + S) PF - P) and feeds a few dozen MFMAs.
//
// Scatter kernels.  Output parity class p = (pz, py, px) of a fine voxel u = 2m + p; per dimension p = 0 takes filter tap 1 at dy[m],
// p = 1 takes tap 2 at dy[m] and tap 0 at dy[m + 1] (conv_trans: one tap per class, no neighbour).  The four waves of a block own the
// four (py, px) classes; a wave runs ALL (dy, dx) neighbour combinations with zero filter fragments where its class has no such tap, so
// every wave executes the same straight-line step (the hand-counted wait needs identical vector-memory counts in every wave).  Along z
// the kernel is input-stationary: coarse plane c finishes fine plane 2c-1 (tap 0), makes fine plane 2c (tap 1) and starts 2c+1 (tap 2).
// The gradient that is already in the destination (the skip tensor's gradient has two writers) and the raw tensor of the norm being
// differentiated travel by LDS-DMA as well, as tiles in the output's own layout: a register load there would be awaited by a
// compiler-placed vmcnt(0) that drains every plane in flight (loads complete in order).  With both, the epilogue also leaves the norm
// backward's statistics {sum g, sum g xhat} (what k_norm_bwd_stats8 computes) -- the accumulating writer sees the COMPLETE gradient.
#include <cstdlib>
#include <type_traits>

#include "mfma_util.h"

namespace unet {

__device__ __attribute__((aligned(16))) unsigned g_s2_zero[4] = {0u, 0u, 0u, 0u};   // what LDS-DMA lanes outside the volume read

struct S2ScatterArgs {
    const void* src;      // coarse tensor [cD][cH][cW][srcC] bf16 (dL/dy of the stride-2 conv; the input of conv_trans)
    int srcC;
    int cD, cH, cW;
    const void* w;        // PK_CONV_S2_DGRAD / PK_CONVT_FWD pack (kernels_mfma_conv.hip)
    const float* bias;    // conv_trans only
    void* out;            // fine tensor [fD][fH][fW][outC] bf16
    int outC;
    int fD, fH, fW;
    int ntt;              // 16-row tiles per parity class in the pack (= outC / 16)
    const void* bn_u = nullptr;        // BNS: raw tensor the gradient belongs to, its norm's {mean, rstd, scale, shift}, partial rows out
    const float* bn_stat = nullptr;
    float* bn_partial = nullptr;
    int bn_act = 0, bn_C = 0;
    int cols_x, cols_y, nseg, zlen;
};

#define S2_DMA(src, dst)                                                                                                     \
    do {                                                                                                                     \
        unsigned keep_;                                                                                                      \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_) : "v"(src), "s"(dst) : "memory");                                                       \
    } while (0)

template <int TR, int KS, bool ACC, bool BNS>
__global__ void __launch_bounds__(256, 2) k_s2_scatter(S2ScatterArgs a) {
    constexpr int BY = 2, BX = 16;
    constexpr int HY = BY + (TR ? 0 : 1), HX = BX + (TR ? 0 : 1);
    constexpr int UPV = KS * 4;                                        // 16-B units per coarse voxel (KS x 32 channels)
    constexpr int IN_UNITS = HY * HX * UPV, IN_UPW = (IN_UNITS + 3) / 4, IN_IT = (IN_UPW + 63) / 64;
    constexpr int IN_B = (4 * IN_UPW * 16 + 255) / 256 * 256;
    constexpr bool IN_FULL = IN_UNITS == 4 * IN_UPW && IN_UPW % 64 == 0;   // every lane of every piece carries a unit
    constexpr int TILE_B = 2 * (2 * BY) * (2 * BX) * 32;               // [2 fine planes][2 BY rows][2 BX voxels][16 channels]
    constexpr int NTILE = (ACC ? 1 : 0) + (BNS ? 1 : 0);
    constexpr int SLOT_B = IN_B + NTILE * TILE_B;
    constexpr int NBUF = 4 * SLOT_B <= 80 * 1024 ? 4 : 3, PF = NBUF - 1;   // two blocks per CU
    constexpr int P = IN_IT + 2 * NTILE, S = 2 * BY;                   // vector-memory operations per wave and step: DMA pieces, stores
    constexpr int WAITN = PF * (P + S) - P;
    static_assert(3 * IN_UPW + (IN_IT - 1) * 64 < IN_UNITS, "every wave issues every piece (the wait counts them)");
    static_assert(TILE_B == 4 * 2 * 64 * 16 && WAITN < 64 && NBUF == PF + 1, "tile pieces / vmcnt range / ring");
    static_assert(!TR || !(ACC || BNS), "conv_trans forward writes");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, gq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), py = wave >> 1, px = wave & 1;
    const int nt0 = blockIdx.y;
    const bf16x8* wp = (const bf16x8*)a.w;
    const bf16x8 zero8 = __builtin_bit_cast(bf16x8, make_uint4(0u, 0u, 0u, 0u));
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    // filter fragments.  TR = 0: wf[role][cy][cx][ks], role 0 = finish plane 2c-1 (pz 1, dz 1), 1 = plane 2c (pz 0), 2 = start plane 2c+1
    // (pz 1, dz 0); zeros where the class has no tap at neighbour (cy, cx).  TR = 1: wf[pz][0][0][ks].
    constexpr int NROLE = TR ? 2 : 3, NC = TR ? 1 : 2;
    bf16x8 wf[NROLE][NC][NC][KS];
#pragma unroll
    for (int role = 0; role < NROLE; ++role)
#pragma unroll
        for (int cy = 0; cy < NC; ++cy)
#pragma unroll
            for (int cx = 0; cx < NC; ++cx)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if constexpr (TR) {
                        const int p = role * 4 + py * 2 + px;
                        wf[role][cy][cx][ks] = wp[((size_t)ks * 8 * a.ntt + (size_t)p * a.ntt + nt0) * 64 + lane];
                    } else {
                        const int pz = role == 1 ? 0 : 1, dz = role == 0 ? 1 : 0;
                        const int p = pz * 4 + py * 2 + px, d = dz * 4 + cy * 2 + cx;
                        const bool have = cy <= py && cx <= px;
                        const bf16x8 v = wp[(((size_t)ks * 8 + d) * 8 * a.ntt + (size_t)p * a.ntt + nt0) * 64 + lane];
                        wf[role][cy][cx][ks] = have ? v : zero8;
                    }
                }
    // patch address of (row 0, neighbour 0, k-step 0) for this lane: voxel j, 16-B channel group gq
    const int mb0 = j * UPV * 16 + gq * 16;

    // staging units of a coarse plane: unit u = (voxel hv = hy HX + hx, 16-B group cg); wave w moves units [w IN_UPW, (w + 1) IN_UPW)
    int iyx[IN_IT];
    bool iact[IN_IT];
    unsigned ipiece[IN_IT];
#pragma unroll
    for (int it = 0; it < IN_IT; ++it) {
        const int u = wave * IN_UPW + it * 64 + lane, hv = u / UPV, cg = u % UPV;
        iact[it] = it * 64 + lane < IN_UPW && u < IN_UNITS;
        iyx[it] = (hv / HX) | ((hv % HX) << 8) | (cg << 16);
        ipiece[it] = (unsigned)__builtin_amdgcn_readfirstlane((wave * IN_UPW + it * 64) * 16);
    }
    // units of an output-layout tile: u = (plane pl, fine row fy, fine x fx, 8-channel half)
    int tyx[2];
    unsigned tpiece[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int u = wave * 128 + it * 64 + lane;
        tyx[it] = ((u >> 6) & 3) | (((u >> 1) & 31) << 8) | ((u >> 8) << 16) | ((u & 1) << 24);
        tpiece[it] = (unsigned)__builtin_amdgcn_readfirstlane((wave * 128 + it * 64) * 16);
    }

    float b4[4], s1[4], s2[4];
    const int cch = nt0 * 16 + gq * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) { b4[r] = (TR && a.bias) ? a.bias[cch + r] : 0.f; s1[r] = 0.f; s2[r] = 0.f; }
    float bmean[4], brstd[4], bsc[4], bsh[4];
    if constexpr (BNS) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            bmean[r] = a.bn_stat[cch + r]; brstd[r] = a.bn_stat[a.bn_C + cch + r];
            bsc[r] = a.bn_stat[2 * a.bn_C + cch + r]; bsh[r] = a.bn_stat[3 * a.bn_C + cch + r];
        }
    }
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)((size_t)a.fD * a.fH * a.fW * a.outC * 2), 0x00020000);
    constexpr int OOB = (int)0x80000000;
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

    const int ncols = a.cols_x * a.cols_y, nitems = ncols * a.nseg;
    for (int item = xcd_remap(blockIdx.x, gridDim.x); item < nitems; item += gridDim.x) {
        const int seg = item / ncols, col = item % ncols;
        const int x0 = (col % a.cols_x) * BX, y0 = (col / a.cols_x) * BY;
        const int zs = seg * a.zlen, ze = zs + a.zlen < a.cD ? zs + a.zlen : a.cD, len = ze - zs;
        const int rlast = TR ? len - 1 : len;             // steps 0 .. rlast: coarse planes zs .. zs + rlast
        const int fz_lo = 2 * zs, fz_hi = 2 * ze < a.fD ? 2 * ze : a.fD;     // fine planes this item stores
        bool iok[IN_IT];
        const char* ibase[IN_IT];
#pragma unroll
        for (int it = 0; it < IN_IT; ++it) {
            const int gy = y0 + (iyx[it] & 255), gx = x0 + ((iyx[it] >> 8) & 255), cg = iyx[it] >> 16;
            iok[it] = iact[it] && gy < a.cH && gx < a.cW;
            ibase[it] = (const char*)a.src + ((size_t)(iok[it] ? gy * a.cW + gx : 0) * a.srcC + cg * 8) * 2;
        }
        const size_t iplane = (size_t)a.cH * a.cW * a.srcC * 2;
        bool tok[2];
        size_t toff[2];     // element offset of the unit's 8 channels inside a fine plane ([y][x][C] with C = outC for both tensors)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int gy = 2 * y0 + (tyx[it] & 255), gx = 2 * x0 + ((tyx[it] >> 8) & 255);
            tok[it] = gy < a.fH && gx < a.fW;
            toff[it] = ((size_t)(tok[it] ? gy * a.fW + gx : 0) * a.outC + nt0 * 16 + (tyx[it] >> 24) * 8);
        }
        const size_t fplane = (size_t)a.fH * a.fW * a.outC;
        auto dma = [&](int r, int slot) {
            const int c = zs + r;
            const bool zin = c < a.cD && r <= rlast;
#pragma unroll
            for (int it = 0; it < IN_IT; ++it) {
                const char* src = (zin && iok[it]) ? ibase[it] + (size_t)c * iplane : (const char*)g_s2_zero;
                const unsigned dst = lds0 + (unsigned)slot * SLOT_B + ipiece[it];
                if (IN_FULL || it + 1 < IN_IT || iact[it]) S2_DMA(src, dst);
            }
            if constexpr (NTILE > 0) {
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int fz = (TR ? 2 * c : 2 * c - 1) + ((tyx[it] >> 16) & 1);
                    const bool ok = r <= rlast && tok[it] && fz >= fz_lo && fz < fz_hi;
                    const size_t e = (size_t)(ok ? fz : 0) * fplane + toff[it];
                    if constexpr (ACC) {
                        const char* src = ok ? (const char*)a.out + e * 2 : (const char*)g_s2_zero;
                        S2_DMA(src, lds0 + (unsigned)slot * SLOT_B + IN_B + tpiece[it]);
                    }
                    if constexpr (BNS) {
                        const char* src = ok ? (const char*)a.bn_u + e * 2 : (const char*)g_s2_zero;
                        S2_DMA(src, lds0 + (unsigned)slot * SLOT_B + IN_B + (ACC ? TILE_B : 0) + tpiece[it]);
                    }
                }
            }
        };
        // output offsets (bytes) of this lane's voxel of row i in fine plane 0
        const int ox = 2 * (x0 + j) + px;
        bool ook[BY];
        unsigned ooff[BY];
#pragma unroll
        for (int i = 0; i < BY; ++i) {
            const int oy = 2 * (y0 + i) + py;
            ook[i] = oy < a.fH && ox < a.fW;
            ooff[i] = (unsigned)((((size_t)(ook[i] ? oy : 0) * a.fW + (ook[i] ? ox : 0)) * a.outC + cch) * 2);
        }
        const unsigned oplane = (unsigned)(fplane * 2);

        f32x4 accC[BY];
#pragma unroll
        for (int i = 0; i < BY; ++i) accC[i] = f32x4{0.f, 0.f, 0.f, 0.f};

        auto compute = [&](int r, auto slc) {
            constexpr int SL = decltype(slc)::value;
            const char* pin = smem + SL * SLOT_B;
            bf16x8 xr[HY][NC][KS];
#pragma unroll
            for (int hy = 0; hy < HY; ++hy)
#pragma unroll
                for (int cx = 0; cx < NC; ++cx)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) xr[hy][cx][ks] = *(const bf16x8*)(pin + mb0 + ((hy * HX + cx) * UPV + ks * 4) * 16);
            const int c = zs + r, zb = TR ? 2 * c : 2 * c - 1;
            auto epilogue = [&](int pl, int i, f32x4 d) {
                const int fz = zb + pl;
                const bool ok = ook[i] && fz >= fz_lo && fz < fz_hi;
                const int off = ok ? (int)(ooff[i] + (unsigned)fz * oplane) : OOB;
                float v0 = d[0] + b4[0], v1 = d[1] + b4[1], v2 = d[2] + b4[2], v3 = d[3] + b4[3];
                const int toffb = ((pl * 2 * BY + 2 * i + py) * (2 * BX) + 2 * j + px) * 32 + gq * 8;
                if constexpr (ACC) {
                    const uint2 old = *(const uint2*)(pin + IN_B + toffb);
                    v0 += bf_lo(old.x); v1 += bf_hi(old.x); v2 += bf_lo(old.y); v3 += bf_hi(old.y);
                }
                u32x2 o;
                o.x = pack_bf16x2(v0, v1); o.y = pack_bf16x2(v2, v3);
                __builtin_amdgcn_raw_buffer_store_b64(o, orsrc, off, 0, 0);
                if constexpr (BNS) {
                    const uint2 uu = *(const uint2*)(pin + IN_B + (ACC ? TILE_B : 0) + toffb);
                    const float uf[4] = {bf_lo(uu.x), bf_hi(uu.x), bf_lo(uu.y), bf_hi(uu.y)};
                    const float rr[4] = {ok ? bf_lo(o.x) : 0.f, ok ? bf_hi(o.x) : 0.f, ok ? bf_lo(o.y) : 0.f, ok ? bf_hi(o.y) : 0.f};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float dv = rr[e] * act_d(fmaf(uf[e], bsc[e], bsh[e]), a.bn_act);
                        s1[e] += dv;
                        s2[e] = fmaf(dv, (uf[e] - bmean[e]) * brstd[e], s2[e]);
                    }
                }
            };
#pragma unroll
            for (int i = 0; i < BY; ++i) {
                if constexpr (TR) {
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) {
                        f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[pl][0][0][ks], xr[i][0][ks], d, 0, 0, 0);
                        epilogue(pl, i, d);
                    }
                } else {
#pragma unroll
                    for (int role = 0; role < 3; ++role) {
                        f32x4 d = role == 0 ? accC[i] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int cy = 0; cy < 2; ++cy)
#pragma unroll
                            for (int cx = 0; cx < 2; ++cx)
#pragma unroll
                                for (int ks = 0; ks < KS; ++ks)
                                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[role][cy][cx][ks], xr[i + cy][cx][ks], d, 0, 0, 0);
                        if (role == 2) accC[i] = d;
                        else epilogue(role, i, d);
                    }
                }
            }
        };
        // step r: wait for coarse plane r (and its tiles), barrier (every wave's pieces have landed; every wave is done with step r - 1,
        // whose slot the request below overwrites: NBUF = PF + 1), request step r + PF, compute
        auto step = [&](int r, auto slc) {
            constexpr int SL = decltype(slc)::value;
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(WAITN) : "memory");
            dma(r + PF, (SL + PF) % NBUF);
            compute(r, slc);
        };
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");     // the previous item's slots are free and its last requests have landed
#pragma unroll
        for (int r = 0; r < PF; ++r) {
            dma(r, r);
#pragma unroll
            for (int i = 0; i < S; ++i)      // out-of-range stores (dropped by the hardware): the prologue's operations count like a step's
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{0u, 0u}, orsrc, OOB + 8 * (r * S + i), 0, 0);
        }
        for (int r = 0; r <= rlast; r += NBUF) {
            step(r, std::integral_constant<int, 0>{});
            if (r + 1 > rlast) break;
            step(r + 1, std::integral_constant<int, 1>{});
            if (r + 2 > rlast) break;
            step(r + 2, std::integral_constant<int, 2>{});
            if constexpr (NBUF == 4) {
                if (r + 3 > rlast) break;
                step(r + 3, std::integral_constant<int, 3>{});
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");    // outstanding requests write LDS: let them land before it is reused
    if constexpr (BNS) {
        float* red = (float*)smem;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float u = s1[r], v = s2[r];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) { u += __shfl_xor(u, m); v += __shfl_xor(v, m); }
            if (j == 0) { red[(wave * 16 + gq * 4 + r) * 2] = u; red[(wave * 16 + gq * 4 + r) * 2 + 1] = v; }
        }
        __syncthreads();
        if (tid < 16) {
            float u = 0.f, v = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { u += red[(w * 16 + tid) * 2]; v += red[(w * 16 + tid) * 2 + 1]; }
            a.bn_partial[((size_t)blockIdx.x * a.bn_C + nt0 * 16 + tid) * 2 + 0] = u;
            a.bn_partial[((size_t)blockIdx.x * a.bn_C + nt0 * 16 + tid) * 2 + 1] = v;
        }
    }
}

// ---- launch ----
static bool s2_off() { return sliding_window_off(); }     // UNET_NO_SLIDING_WINDOW: k_mfma_conv_p for every shape
static void s2_work(S2ScatterArgs& a, int gy, int* gx) {
    a.cols_x = (a.cW + 15) / 16; a.cols_y = (a.cH + 1) / 2;
    const int cols = a.cols_x * a.cols_y;
    int want = 512 / gy;
    if (want < 1) want = 1;
    int nseg = (want + cols - 1) / cols;
    if (nseg < 1) nseg = 1;
    int zlen = (a.cD + nseg - 1) / nseg;
    if (zlen < 2) zlen = 2;
    if (zlen > a.cD) zlen = a.cD;
    a.nseg = (a.cD + zlen - 1) / zlen; a.zlen = zlen;
    const int items = cols * a.nseg;
    *gx = items < want ? items : want;
}
template <int TR, int KS, bool ACC, bool BNS>
static void launch_scatter_t(const S2ScatterArgs& a, int gx, int gy, hipStream_t s) {
    constexpr int HY = 2 + (TR ? 0 : 1), HX = 16 + (TR ? 0 : 1), IN_UNITS = HY * HX * KS * 4, IN_UPW = (IN_UNITS + 3) / 4;
    constexpr int IN_B = (4 * IN_UPW * 16 + 255) / 256 * 256;
    constexpr int SLOT_B = IN_B + ((ACC ? 1 : 0) + (BNS ? 1 : 0)) * 8192, lds = (4 * SLOT_B <= 80 * 1024 ? 4 : 3) * SLOT_B;
    static_assert(lds <= 80 * 1024, "two blocks per CU");
    static std::atomic<uint64_t> attr_done{0};
    set_max_lds_once(attr_done, (const void*)k_s2_scatter<TR, KS, ACC, BNS>, lds);
    k_s2_scatter<TR, KS, ACC, BNS><<<dim3((unsigned)gx, (unsigned)gy), 256, lds, s>>>(a);
}

// Conv3d(k3, s2) input gradient (g = forward geometry).  Returns 0 when the shape is not served (the caller uses k_mfma_conv_p),
// else the number of norm-backward partial rows written when bn was given (the grid's x size), or -1 when it ran without statistics.
int launch_s2_conv_dgrad(const ConvGeom& g, const void* dy, const void* w_s2_dgrad, const DstGrad* dst, int ndst, hipStream_t s,
                         const BnBwdStats* bn) {
    if (s2_off() || g.ks != 3 || g.stride != 2 || ndst != 1 || !dst[0].ptr || dst[0].C != g.Cin) return 0;
    if (g.Cin % 16 || (g.Cout != 32 && g.Cout != 64) || g.Wo < 16 || g.Do < 4) return 0;
    if ((size_t)g.D * g.H * g.W * g.Cin * 2 >= ((size_t)1 << 31)) return 0;       // 31-bit buffer offsets
    S2ScatterArgs a;
    a.src = dy; a.srcC = g.Cout; a.cD = g.Do; a.cH = g.Ho; a.cW = g.Wo;
    a.w = w_s2_dgrad; a.bias = nullptr;
    a.out = dst[0].ptr; a.outC = g.Cin; a.fD = g.D; a.fH = g.H; a.fW = g.W; a.ntt = g.Cin / 16;
    const bool acc = dst[0].accumulate != 0;
    const bool bns = bn && bn->partial && bn->C == g.Cin;
    if (bns) { a.bn_u = bn->u; a.bn_stat = bn->stat; a.bn_partial = bn->partial; a.bn_act = bn->act; a.bn_C = bn->C; }
    int gx = 0;
    const int gy = g.Cin / 16;
    s2_work(a, gy, &gx);
    const int ks = g.Cout / 32;
#define S2_GO(KS_)                                                                   \
    do {                                                                             \
        if (acc && bns) launch_scatter_t<0, KS_, true, true>(a, gx, gy, s);          \
        else if (acc) launch_scatter_t<0, KS_, true, false>(a, gx, gy, s);           \
        else if (bns) launch_scatter_t<0, KS_, false, true>(a, gx, gy, s);           \
        else launch_scatter_t<0, KS_, false, false>(a, gx, gy, s);                   \
    } while (0)
    if (ks == 1) S2_GO(1); else S2_GO(2);
#undef S2_GO
    return bns ? gx : -1;
}
int s2_conv_dgrad_rows_max() { return 512; }

// ConvTranspose3d(k2, s2) forward; false when the shape is not served
bool launch_s2_convt_fwd(const ConvGeom& g, const SrcDesc* src, int nsrc, const void* w_mfma, const float* bias, void* out, hipStream_t s) {
    if (s2_off() || nsrc != 1 || src[0].scale || src[0].act || src[0].C != g.Cin) return false;
    if ((g.Cin != 32 && g.Cin != 64 && g.Cin != 128) || g.Cout % 16 || g.W < 16 || g.D < 4) return false;
    if ((size_t)g.Do * g.Ho * g.Wo * g.Cout * 2 >= ((size_t)1 << 31)) return false;
    S2ScatterArgs a;
    a.src = src[0].ptr; a.srcC = g.Cin; a.cD = g.D; a.cH = g.H; a.cW = g.W;
    a.w = w_mfma; a.bias = bias;
    a.out = out; a.outC = g.Cout; a.fD = g.Do; a.fH = g.Ho; a.fW = g.Wo; a.ntt = g.Cout / 16;
    int gx = 0;
    const int gy = g.Cout / 16;
    s2_work(a, gy, &gx);
    if (g.Cin == 32) launch_scatter_t<1, 1, false, false>(a, gx, gy, s);
    else if (g.Cin == 64) launch_scatter_t<1, 2, false, false>(a, gx, gy, s);
    else launch_scatter_t<1, 4, false, false>(a, gx, gy, s);
    return true;
}

// ------------------------------------------------------------------------------------------------
// Gather kernels (fine -> coarse): Conv3d(k3, s2, p1) forward and ConvTranspose3d(k2, s2) input gradient.
// A block owns 4 x 16 output voxels in (y, x) and walks z over a segment of output planes; each step brings ONE fine input plane
// (9 x 33 voxels for k3, 8 x 32 for k2) by LDS-DMA.  The plane's LDS image is space-to-depth in x: a row holds its odd-x voxels
// first, then its even-x voxels (k2: even, then odd), so the 16 output voxels of an MFMA column read 16 CONSECUTIVE voxels for every
// tap (the same conflict-free pattern as k_mfma_conv_z16; the halo-tile kernel's stride-2 reads spent a third of the LDS cycles on
// bank conflicts).  The permutation costs nothing: an LDS-DMA's LDS image is lane-linear, so it is applied to the SOURCE address.
// Input-stationary along z: an odd fine plane 2m+1 feeds output planes m (tap kz = 2) and m+1 (kz = 0), an even one output plane m
// (kz = 1); k2: plane 2m+a feeds output plane m.  Waves: WM row tiles x (4 / WM) groups of WM output rows; one row tile per wave.
// Epilogue as k_mfma_conv_z16: bias, bf16, 8-B stores through a buffer descriptor, per-thread norm statistics -> one row per block.
// ------------------------------------------------------------------------------------------------
struct S2GatherArgs {
    SrcDesc src[2];       // fine tensor(s) [fD][fH][fW][C] bf16, plain (activated copies); a channel concat is a second pointer
    int nsrc;
    int fD, fH, fW;
    const void* w;        // PK_CONV_FWD (16-channel chunks, 27 taps) / PK_CONVT_DGRAD (16-channel chunks, 8 taps) pack
    const float* bias;
    void* out;            // coarse tensor [cD][cH][cW][outC] bf16
    int outC;
    int cD, cH, cW;
    float* stats;         // [gridDim.x][outC][2] or nullptr
    int cols_x, cols_y, nseg, zlen;
};

template <int KS, int CIN, int WM>
__global__ void __launch_bounds__(256, 2) k_s2_gather(S2GatherArgs a) {
    constexpr int BYO = 4, BX = 16, NR = WM;                            // NR output rows per wave
    constexpr int HY = 2 * BYO + (KS == 3 ? 1 : 0), NODD = KS == 3 ? 17 : 16, HXV = NODD + 16;
    constexpr int UPV = CIN / 8, VB = CIN * 2;
    constexpr int UNITS = HY * HXV * UPV, UPW = (UNITS + 3) / 4, IT = (UPW + 63) / 64;
    constexpr int PLANE_B = (4 * UPW * 16 + 255) / 256 * 256;
    constexpr bool FULL = UNITS == 4 * UPW && UPW % 64 == 0;          // every lane of every piece carries a unit
    constexpr int NBUF = CIN == 16 ? 6 : 4, PF = NBUF - 1;
    constexpr int NT2 = KS * KS, NKS = CIN == 16 ? (NT2 + 1) / 2 : NT2; // (ky, kx) taps of a plane; k-steps of a plane
    constexpr int KT = (KS * KS * KS + 1) / 2;                          // k-steps per 16-channel chunk in the pack
    constexpr int P = IT, S = NR, WAITN = PF * (P + S) - P;
    static_assert(3 * UPW + (IT - 1) * 64 < UNITS && WAITN < 64 && NBUF % 2 == 0, "pieces / vmcnt range / plane parity = slot parity");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, gq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave % WM, wr = wave / WM;
    const int nt = blockIdx.y * WM + wm, NTT = a.outC / 16, C0 = a.src[0].C;
    const bf16x8* wp = (const bf16x8*)a.w;
    const bf16x8 zero8 = __builtin_bit_cast(bf16x8, make_uint4(0u, 0u, 0u, 0u));
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    // filter fragments wf[kz][s] and the patch address of k-step s for this lane's output row 0
    bf16x8 wf[KS][NKS];
    int mb[NKS];
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
        const int T = CIN == 16 ? 2 * s + (gq >> 1) : s;               // this lane's (ky, kx) tap of the k-step
        const int Tc = T < NT2 ? T : NT2 - 1, ky = Tc / KS, kx = Tc % KS;
        const int sb = KS == 3 ? (kx == 0 ? 0 : (kx == 1 ? NODD : 1)) : (kx == 0 ? 0 : NODD);
        mb[s] = ((wr * NR * 2 + ky) * HXV + sb + j) * VB + (CIN == 16 ? (gq & 1) : gq) * 16;
#pragma unroll
        for (int kz = 0; kz < KS; ++kz) {
            const int tg = kz * NT2 + Tc, q = CIN == 16 ? 0 : (gq >> 1), half = gq & 1;
            const bf16x8 v = wp[(((size_t)q * KT + (tg >> 1)) * NTT + nt) * 64 + ((lane & 15) | (half << 4) | ((tg & 1) << 5))];
            wf[kz][s] = T < NT2 ? v : zero8;
        }
    }
    // staging units: unit u = (voxel hv = hy HXV + slot, 16-B group); slot -> fine x: the odd-x voxels first (k2: the even ones)
    int uyx[IT];
    bool uact[IT];
    unsigned upiece[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int u = wave * UPW + it * 64 + lane, hv = u / UPV, hy = hv / HXV, sl = hv % HXV;
        const int t = sl < NODD ? 2 * sl : 2 * (sl - NODD) + 1;        // x = 2 x0 - (KS == 3) + t
        uact[it] = it * 64 + lane < UPW && u < UNITS;
        uyx[it] = hy | (t << 8) | ((u % UPV) << 16);
        upiece[it] = (unsigned)__builtin_amdgcn_readfirstlane((wave * UPW + it * 64) * 16);
    }
    float b4[4], s1[4], s2[4];
    const int cch = nt * 16 + gq * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) { b4[r] = a.bias ? a.bias[cch + r] : 0.f; s1[r] = 0.f; s2[r] = 0.f; }
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)((size_t)a.cD * a.cH * a.cW * a.outC * 2), 0x00020000);
    constexpr int OOB = (int)0x80000000;
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

    const int ncols = a.cols_x * a.cols_y, nitems = ncols * a.nseg;
    for (int item = xcd_remap(blockIdx.x, gridDim.x); item < nitems; item += gridDim.x) {
        const int seg = item / ncols, col = item % ncols;
        const int x0 = (col % a.cols_x) * BX, y0 = (col / a.cols_x) * BYO;
        const int zs = seg * a.zlen, ze = zs + a.zlen < a.cD ? zs + a.zlen : a.cD, len = ze - zs;
        const int rlast = KS == 3 ? 2 * len : 2 * len - 1;              // steps 0 .. rlast: fine planes pz0 .. pz0 + rlast
        const int pz0 = 2 * zs - (KS == 3 ? 1 : 0);
        bool uok[IT];
        const char* ubase[IT];
        unsigned uvs[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int gy = 2 * y0 - (KS == 3 ? 1 : 0) + (uyx[it] & 255), gx = 2 * x0 - (KS == 3 ? 1 : 0) + ((uyx[it] >> 8) & 255);
            const int c = (uyx[it] >> 16) * 8, sidx = (a.nsrc > 1 && c >= C0) ? 1 : 0;
            uok[it] = uact[it] && (unsigned)gy < (unsigned)a.fH && (unsigned)gx < (unsigned)a.fW;
            uvs[it] = (unsigned)(sidx ? a.src[1].C : C0) * 2;
            ubase[it] = (const char*)(sidx ? a.src[1].ptr : a.src[0].ptr) + (size_t)(c - (sidx ? C0 : 0)) * 2 +
                        (size_t)(uok[it] ? gy * a.fW + gx : 0) * uvs[it];
        }
        const size_t hw = (size_t)a.fH * a.fW;
        auto dma = [&](int r, int slot) {
            const int pz = pz0 + r;
            const bool zin = (unsigned)pz < (unsigned)a.fD && r <= rlast;
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const char* src = (zin && uok[it]) ? ubase[it] + (size_t)pz * hw * uvs[it] : (const char*)g_s2_zero;
                const unsigned dst = lds0 + (unsigned)slot * PLANE_B + upiece[it];
                if (FULL || it + 1 < IT || uact[it]) S2_DMA(src, dst);
            }
        };
        const int ox = x0 + j;
        bool ook[NR];
        unsigned ooff[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int oy = y0 + wr * NR + i;
            ook[i] = oy < a.cH && ox < a.cW;
            ooff[i] = (unsigned)(((((size_t)zs * a.cH + (ook[i] ? oy : 0)) * a.cW + (ook[i] ? ox : 0)) * a.outC + cch) * 2);
        }
        const unsigned oplane = (unsigned)((size_t)a.cH * a.cW * a.outC * 2);
        f32x4 accC[NR], accN[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) { accC[i] = f32x4{0.f, 0.f, 0.f, 0.f}; accN[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }

        auto compute = [&](int r, auto slc) {
            constexpr int SL = decltype(slc)::value, PAR = SL & 1;    // r and its slot have the same parity (NBUF is even)
            const char* pin = smem + SL * PLANE_B;
            // which planes complete here: k3: an odd fine plane (PAR 0) completes output plane r / 2 - 1; k2: fine plane 2m + 1 (PAR 1)
            constexpr bool FIN = KS == 3 ? PAR == 0 : PAR == 1;
#pragma unroll
            for (int i = 0; i < NR; ++i)
#pragma unroll
                for (int s = 0; s < NKS; ++s) {
                    const bf16x8 x = *(const bf16x8*)(pin + mb[s] + i * 2 * HXV * VB);
                    if constexpr (KS == 3 && PAR == 0) {
                        accC[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[2][s], x, accC[i], 0, 0, 0);
                        accN[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][s], x, s == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : accN[i], 0, 0, 0);
                    } else if constexpr (KS == 3) {
                        accC[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][s], x, accC[i], 0, 0, 0);
                    } else if constexpr (PAR == 0) {
                        accC[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][s], x, s == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : accC[i], 0, 0, 0);
                    } else {
                        accC[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][s], x, accC[i], 0, 0, 0);
                    }
                    // one row's fragments in flight at a time where a wave owns four rows (left alone the scheduler hoists all 36 reads
                    // of a 32-channel plane above the first MFMA: 144 VGPRs on top of 27 filter fragments -> scratch inside a counted step)
                    if constexpr (NR * NKS > 18) { if (s == NKS - 1) __builtin_amdgcn_sched_barrier(0); }
                }
            const int k = KS == 3 ? (r >> 1) - 1 : (r >> 1);
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                if constexpr (FIN) {
                    const f32x4 d = accC[i];
                    const bool ok = k >= 0 && k < len && ook[i];
                    const int off = ok ? (int)(ooff[i] + (unsigned)k * oplane) : OOB;
                    u32x2 o;
                    o.x = pack_bf16x2(d[0] + b4[0], d[1] + b4[1]); o.y = pack_bf16x2(d[2] + b4[2], d[3] + b4[3]);
                    __builtin_amdgcn_raw_buffer_store_b64(o, orsrc, off, 0, 0);
                    const float r0 = ok ? bf_lo(o.x) : 0.f, r1 = ok ? bf_hi(o.x) : 0.f, r2 = ok ? bf_lo(o.y) : 0.f, r3 = ok ? bf_hi(o.y) : 0.f;
                    s1[0] += r0; s1[1] += r1; s1[2] += r2; s1[3] += r3;
                    s2[0] = fmaf(r0, r0, s2[0]); s2[1] = fmaf(r1, r1, s2[1]); s2[2] = fmaf(r2, r2, s2[2]); s2[3] = fmaf(r3, r3, s2[3]);
                    if constexpr (KS == 3) accC[i] = accN[i];
                } else {
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{0u, 0u}, orsrc, OOB + 8 * (i + 1), 0, 0);    // keeps the step's store count
                }
            }
        };
        auto step = [&](int r, auto slc) {
            constexpr int SL = decltype(slc)::value;
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(WAITN) : "memory");
            dma(r + PF, (SL + PF) % NBUF);
            compute(r, slc);
        };
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int r = 0; r < PF; ++r) {
            dma(r, r);
#pragma unroll
            for (int i = 0; i < S; ++i) __builtin_amdgcn_raw_buffer_store_b64(u32x2{0u, 0u}, orsrc, OOB + 64 + 8 * (r * S + i), 0, 0);
        }
        for (int r = 0; r <= rlast; r += NBUF) {
            step(r, std::integral_constant<int, 0>{});
            if (r + 1 > rlast) break;
            step(r + 1, std::integral_constant<int, 1>{});
            if (r + 2 > rlast) break;
            step(r + 2, std::integral_constant<int, 2>{});
            if (r + 3 > rlast) break;
            step(r + 3, std::integral_constant<int, 3>{});
            if constexpr (NBUF == 6) {
                if (r + 4 > rlast) break;
                step(r + 4, std::integral_constant<int, 4>{});
                if (r + 5 > rlast) break;
                step(r + 5, std::integral_constant<int, 5>{});
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (a.stats) {
        float* red = (float*)smem;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float u = s1[r], v = s2[r];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) { u += __shfl_xor(u, m); v += __shfl_xor(v, m); }
            if (j == 0) { red[(wave * 16 + gq * 4 + r) * 2] = u; red[(wave * 16 + gq * 4 + r) * 2 + 1] = v; }
        }
        __syncthreads();
        if (tid < WM * 16) {       // channel tid of this block's WM row tiles: the sums of the 4 / WM waves that own it
            const int m = tid >> 4, c = tid & 15;
            float u = 0.f, v = 0.f;
#pragma unroll
            for (int g = 0; g < 4 / WM; ++g) { u += red[((g * WM + m) * 16 + c) * 2]; v += red[((g * WM + m) * 16 + c) * 2 + 1]; }
            a.stats[((size_t)blockIdx.x * a.outC + (blockIdx.y * WM + m) * 16 + c) * 2 + 0] = u;
            a.stats[((size_t)blockIdx.x * a.outC + (blockIdx.y * WM + m) * 16 + c) * 2 + 1] = v;
        }
    }
}

static void s2_gather_work(S2GatherArgs& a, int gy, int* gx) {
    a.cols_x = (a.cW + 15) / 16; a.cols_y = (a.cH + 3) / 4;
    const int cols = a.cols_x * a.cols_y;
    int want = 512 / gy;
    if (want < 1) want = 1;
    int nseg = (want + cols - 1) / cols;
    if (nseg < 1) nseg = 1;
    int zlen = (a.cD + nseg - 1) / nseg;
    if (zlen < 2) zlen = 2;
    if (zlen > a.cD) zlen = a.cD;
    a.nseg = (a.cD + zlen - 1) / zlen; a.zlen = zlen;
    const int items = cols * a.nseg;
    *gx = items < want ? items : want;
}
template <int KS, int CIN, int WM>
static void launch_gather_t(const S2GatherArgs& a, int gx, int gy, hipStream_t s) {
    constexpr int HY = 8 + (KS == 3 ? 1 : 0), HXV = (KS == 3 ? 17 : 16) + 16, UNITS = HY * HXV * (CIN / 8), UPW = (UNITS + 3) / 4;
    constexpr int PLANE_B = (4 * UPW * 16 + 255) / 256 * 256, lds = (CIN == 16 ? 6 : 4) * PLANE_B;
    static_assert(lds <= 80 * 1024, "two blocks per CU");
    static std::atomic<uint64_t> attr_done{0};
    set_max_lds_once(attr_done, (const void*)k_s2_gather<KS, CIN, WM>, lds);
    k_s2_gather<KS, CIN, WM><<<dim3((unsigned)gx, (unsigned)gy), 256, lds, s>>>(a);
}
template <int KS>
static void launch_gather(const S2GatherArgs& a, int cin, int wm, int gx, int gy, hipStream_t s) {
    if (cin == 16) {
        if (wm == 4) launch_gather_t<KS, 16, 4>(a, gx, gy, s); else if (wm == 2) launch_gather_t<KS, 16, 2>(a, gx, gy, s); else launch_gather_t<KS, 16, 1>(a, gx, gy, s);
    } else {
        if constexpr (KS == 2) { if (wm == 4) { launch_gather_t<KS, 32, 4>(a, gx, gy, s); return; } }
        if (wm == 2) launch_gather_t<KS, 32, 2>(a, gx, gy, s); else launch_gather_t<KS, 32, 1>(a, gx, gy, s);
    }
}
// row tiles per block (= waves along the rows).  k3 with 32-channel planes: at most two -- a wave that owns four output rows keeps 27
// filter fragments + two accumulator planes of four rows and spilled into scratch (a vector-memory operation inside a counted step)
static int s2_wm(int rows, int ks, int cin) {
    const int t = rows / 16;
    int wm = t % 4 == 0 ? 4 : (t % 2 == 0 ? 2 : 1);
    if (ks == 3 && cin == 32 && wm == 4) wm = 2;
    return wm;
}

// Conv3d(k3, s2) forward; returns 0 when the shape is not served, else the number of statistics rows (the grid's x size)
int launch_s2_conv_fwd(const ConvGeom& g, const SrcDesc* src, int nsrc, const void* w_mfma, const float* bias, void* out, float* stats_partial,
                       hipStream_t s) {
    if (s2_off() || g.ks != 3 || g.stride != 2 || (g.Cin != 16 && g.Cin != 32) || g.Cout % 16 || g.Wo < 16 || g.Do < 4) return 0;
    int csum = 0;
    for (int k = 0; k < nsrc; ++k) { if (src[k].C % 8 || src[k].scale || src[k].act) return 0; csum += src[k].C; }
    if (csum != g.Cin || (size_t)g.Do * g.Ho * g.Wo * g.Cout * 2 >= ((size_t)1 << 31)) return 0;
    S2GatherArgs a;
    a.nsrc = nsrc; a.src[0] = src[0]; if (nsrc > 1) a.src[1] = src[1];
    a.fD = g.D; a.fH = g.H; a.fW = g.W;
    a.w = w_mfma; a.bias = bias; a.out = out; a.outC = g.Cout; a.cD = g.Do; a.cH = g.Ho; a.cW = g.Wo; a.stats = stats_partial;
    const int wm = s2_wm(g.Cout, 3, g.Cin), gy = g.Cout / (16 * wm);
    int gx = 0;
    s2_gather_work(a, gy, &gx);
    launch_gather<3>(a, g.Cin, wm, gx, gy, s);
    return gx;
}
// ConvTranspose3d(k2, s2) input gradient (g = forward geometry of the conv_trans); false when the shape is not served
bool launch_s2_convt_dgrad(const ConvGeom& g, const void* dy, const void* w_mfma_dgrad, const DstGrad* dst, int ndst, hipStream_t s) {
    if (s2_off() || ndst != 1 || !dst[0].ptr || dst[0].accumulate || dst[0].C != g.Cin) return false;
    if ((g.Cout != 16 && g.Cout != 32) || g.Cin % 16 || g.W < 16 || g.D < 4) return false;
    if ((size_t)g.D * g.H * g.W * g.Cin * 2 >= ((size_t)1 << 31)) return false;
    S2GatherArgs a;
    a.nsrc = 1; a.src[0].ptr = dy; a.src[0].C = g.Cout;
    a.fD = g.Do; a.fH = g.Ho; a.fW = g.Wo;
    a.w = w_mfma_dgrad; a.bias = nullptr; a.out = dst[0].ptr; a.outC = g.Cin; a.cD = g.D; a.cH = g.H; a.cW = g.W; a.stats = nullptr;
    const int wm = s2_wm(g.Cin, 2, g.Cout), gy = g.Cin / (16 * wm);
    int gx = 0;
    s2_gather_work(a, gy, &gx);
    launch_gather<2>(a, g.Cout, wm, gx, gy, s);
    return true;
}

}  // namespace unet
