// k_mfma_wgrad_z (kernels_mfma_wgrad_z.hip) with its planes brought by LDS-DMA: the sliding-window weight gradient of the 3x3x3 stride-1
// convolutions,  dW[co][ci][kz][ky][kx] = sum_v a[v + (kz-1, ky-1, kx-1)][ci] * dy[v][co]   (autograd of unet.cpp:59-72, train.cpp:706).
//
// Same decomposition and the same MFMA schedule as k_mfma_wgrad_z -- a block owns a (BY x 32) footprint and walks z; the dy fragments of
// the three planes z-1, z, z+1 rotate through registers so that one transposing A-fragment read pair feeds up to 9 MFMAs; 27 tap
// accumulators per wave; waves split rows (WK) and (ca, cb) tile pairs (PA x PB); slab per block + fixed-order reduce -- but the input
// plane (halo) and the dy plane of a step travel HBM -> LDS by global_load_lds_dwordx4, PF = 3 steps ahead into a ring of four buffers:
//   * no staging registers (k_mfma_wgrad_z holds two or three register sets of 16-B loads: 40 of its 237 VGPRs) and no ds_write_b128
//     pass (LDS stores run at ~79 B/clk/CU: at 17 KB per step they cost as much as the step's MFMAs) -- PMC of the register-staged kernel
//     (profiles/r14_wgrad_polite_32to16_128_counters.txt): MFMA pipe 32 % busy, LDS 24 %, HBM 3.0 TB/s, i.e. phases that do not overlap;
//   * three steps of loads in flight per block instead of two;
//   * the loop's only vector-memory operations are the DMA pieces, the same number in every wave and step (P), so a plane is awaited
//     with ONE hand-counted s_waitcnt vmcnt((PF - 1) P) (checked on the emitted code at build time: tools/check_asm_loads.py s2dma rules).
// Used for the 4-wave configurations (the polite launches of the engine's side stream and the <= 32^3 volumes).
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "mfma_util.h"

namespace unet {

__device__ __attribute__((aligned(16))) unsigned g_wzd_zero[4] = {0u, 0u, 0u, 0u};

typedef __attribute__((ext_vector_type(4))) short zd16x4;
typedef __attribute__((address_space(3))) zd16x4 zdlds_s16x4;
__device__ __forceinline__ bf16x8 zdtr_read2(const char* p0, const char* p1) {
    zd16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((zdlds_s16x4*)p0);
    zd16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((zdlds_s16x4*)p1);
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

struct WgradZdArgs {
    ConvGeom g;        // Cin = Ca (input channels), Cout = Cb (dy channels); D,H,W = volume
    SrcDesc asrc[2];   // input (may be a channel concat), plain
    int nasrc;
    const void* dy;
    float* slab;       // [gridDim.x][Cb][Ca][27]
    float* bias_slab;  // [gridDim.x][Cb] or nullptr
    int cols_x, cols_y, nseg, zlen;
};

#define WZD_DMA(src, dst)                                                                                                    \
    do {                                                                                                                     \
        unsigned keep_;                                                                                                      \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_) : "v"(src), "s"(__builtin_amdgcn_readfirstlane((int)(dst))) : "memory");                 \
    } while (0)

template <int WK, int PA, int PB>
__global__ void __launch_bounds__(256, 2) k_mfma_wgrad_zd(WgradZdArgs a) {
    constexpr int P = PA * PB, T = 27, BX = 32, R = 2, BY = R * WK, HY = BY + 2, HX = BX + 2;
    static_assert(WK * P == 4, "four waves");
    constexpr int NVA = HY * HX, NVB = BY * BX;
    constexpr int APLANE = NVA * 32, BPLANE = NVB * 32, BUF = PA * APLANE + PB * BPLANE;
    constexpr int NB = 4, PF = 3;
    constexpr int UA = PA * NVA * 2, UB = PB * NVB * 2;                 // 16-B units of a step; LDS image [tile][voxel][half] is linear in the unit index
    constexpr int UPWA = (UA + 3) / 4, UPWB = (UB + 3) / 4, ITA = (UPWA + 63) / 64, ITB = (UPWB + 63) / 64;
    constexpr int PP = ITA + ITB, WAITN = (PF - 1) * PP;                // DMA pieces per wave and step; the hand-counted wait
    static_assert(3 * UPWA + (ITA - 1) * 64 < UA && 3 * UPWB + (ITB - 1) * 64 < UB, "every wave issues every piece");
    static_assert(UPWB % 2 == 0 && (PB == 1 || (NVB * 2) % UPWB == 0), "a thread's dy units are all of one (tile, half)");
    constexpr bool FULLA = UA == 4 * UPWA && UPWA % 64 == 0, FULLB = UB == 4 * UPWB && UPWB % 64 == 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, il = lane & 15, gq = lane >> 4, q4 = il >> 2, p4 = il & 3;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), pw = wave % P, kw = wave / P, ia = pw / PB, ib = pw % PB;
    const int CBG = (g.Cout / 16) / PB;
    const int caB = ((int)blockIdx.y / CBG) * PA, cbB = ((int)blockIdx.y % CBG) * PB;
    const int C0 = a.asrc[0].C;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    const int item = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int ncols = a.cols_x * a.cols_y, seg = item / ncols, col = item % ncols;
    const int x0 = (col % a.cols_x) * BX, y0 = (col / a.cols_x) * BY;
    const int zs = seg * a.zlen, ze = zs + a.zlen < g.D ? zs + a.zlen : g.D, len = ze - zs;

    f32x4 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum[e] = 0.f;
    const bool do_bias = a.bias_slab != nullptr && caB == 0;

    // ---- staging units, fixed per thread: source pointer of the unit in plane 0 (nullptr: outside the volume in (y, x)) ----
    const char* abase[ITA];
    const char* bbase[ITB];
    size_t aplane[ITA];
    bool aact[ITA], bact[ITB];
    unsigned apiece[ITA], bpiece[ITB];
#pragma unroll
    for (int it = 0; it < ITA; ++it) {
        const int u = wave * UPWA + it * 64 + lane, tile = u / (NVA * 2), hv = (u >> 1) % NVA, half = u & 1, hy = hv / HX, hx = hv % HX;
        const int gy = y0 - 1 + hy, gx = x0 - 1 + hx, c = (caB + tile) * 16 + half * 8;
        const int sa = (a.nasrc > 1 && c >= C0) ? 1 : 0, aC = sa ? a.asrc[1].C : C0;
        aact[it] = it * 64 + lane < UPWA && u < UA;
        const bool ok = aact[it] && (unsigned)gy < (unsigned)g.H && (unsigned)gx < (unsigned)g.W;
        abase[it] = ok ? (const char*)(sa ? a.asrc[1].ptr : a.asrc[0].ptr) + (((size_t)gy * g.W + gx) * aC + (c - (sa ? C0 : 0))) * 2 : nullptr;
        aplane[it] = (size_t)g.H * g.W * aC * 2;
        apiece[it] = (unsigned)__builtin_amdgcn_readfirstlane((wave * UPWA + it * 64) * 16);
    }
#pragma unroll
    for (int it = 0; it < ITB; ++it) {
        const int u = wave * UPWB + it * 64 + lane, tile = u / (NVB * 2), tv = (u >> 1) % NVB, half = u & 1, ty = tv / BX, tx = tv % BX;
        const int gy = y0 + ty, gx = x0 + tx;
        bact[it] = it * 64 + lane < UPWB && u < UB;
        const bool ok = bact[it] && gy < g.H && gx < g.W;
        bbase[it] = ok ? (const char*)a.dy + (((size_t)gy * g.W + gx) * g.Cout + (cbB + tile) * 16 + half * 8) * 2 : nullptr;
        bpiece[it] = (unsigned)__builtin_amdgcn_readfirstlane((PA * APLANE + (wave * UPWB + it * 64) * 16));
    }
    const size_t bplane = (size_t)g.H * g.W * g.Cout * 2;
    // step k (k = 0 .. len + 1): input plane zs - 1 + k, dy plane zs + k (k < len) -> ring buffer k % NB.  Planes outside the volume /
    // the segment read the zero page; every wave issues exactly PP pieces per request.
    auto request = [&](int k) {
        const int pz = zs - 1 + k, bz = zs + k;
        const bool ain = (unsigned)pz < (unsigned)g.D && k <= len + 1, bin = bz < ze;
        const unsigned buf = lds0 + (unsigned)(k % NB) * BUF;
#pragma unroll
        for (int it = 0; it < ITA; ++it) {
            const char* src = (ain && abase[it]) ? abase[it] + (size_t)pz * aplane[it] : (const char*)g_wzd_zero;
            if (FULLA || it + 1 < ITA || aact[it]) WZD_DMA(src, buf + apiece[it]);
        }
#pragma unroll
        for (int it = 0; it < ITB; ++it) {
            const char* src = (bin && bbase[it]) ? bbase[it] + (size_t)bz * bplane : (const char*)g_wzd_zero;
            if (FULLB || it + 1 < ITB || bact[it]) WZD_DMA(src, buf + bpiece[it]);
        }
    };

    // ---- fragment addresses (as k_mfma_wgrad_z): lane group gq, read r fetch voxel group G = gq + 4r of the K-step ----
    int aoff[2], boff[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int x = (gq + 4 * r) * 4 + q4;
        aoff[r] = ia * APLANE + ((kw * R) * HX + x) * 32 + p4 * 8;
        boff[r] = PA * APLANE + ib * BPLANE + ((kw * R) * BX + x) * 32 + p4 * 8;
    }
    const bf16x8 zero8 = __builtin_bit_cast(bf16x8, make_uint4(0u, 0u, 0u, 0u));
    bf16x8 Bq[3][R];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int j = 0; j < R; ++j) Bq[s][j] = zero8;

    // one input plane: PH = register slot of the NEW dy plane (kz = 0); kz = 1 -> slot PH + 2, kz = 2 -> slot PH + 1 (mod 3)
    auto compute = [&](auto ph, const char* buf, bool bnew) {
        constexpr int PH = decltype(ph)::value;
        const char* pa0 = buf + aoff[0];
        const char* pa1 = buf + aoff[1];
        const char* pb0 = buf + boff[0];
        const char* pb1 = buf + boff[1];
#pragma unroll
        for (int j = 0; j < R; ++j) Bq[PH][j] = bnew ? zdtr_read2(pb0 + j * BX * 32, pb1 + j * BX * 32) : zero8;
        constexpr int NA = R + 2, NREAD = NA * 3, RD = 4;      // A fragments: rows 0 .. R + 1 of this wave's strip x 3 shifts
        bf16x8 ring[RD];
#pragma unroll
        for (int i = 0; i < RD && i < NREAD; ++i) ring[i] = zdtr_read2(pa0 + ((i / 3) * HX + i % 3) * 32, pa1 + ((i / 3) * HX + i % 3) * 32);
#pragma unroll
        for (int i = 0; i < NREAD; ++i) {
            const int ai = i / 3, kx = i % 3;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kz = 2; kz >= 0; --kz)
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    const int ky = ai - j;
                    if (ky >= 0 && ky <= 2)
                        acc[kz * 9 + ky * 3 + kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[i % RD], Bq[(PH + (3 - kz)) % 3][j], acc[kz * 9 + ky * 3 + kx], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
            if (i + RD < NREAD) {
                const int m = i + RD;
                ring[m % RD] = zdtr_read2(pa0 + ((m / 3) * HX + m % 3) * 32, pa1 + ((m / 3) * HX + m % 3) * 32);
            }
        }
    };
    // bias: the dy units this thread requested (all of one (tile, half)), read back from LDS once they have landed
    auto bias_add = [&](const char* buf) {
#pragma unroll
        for (int it = 0; it < ITB; ++it) {
            if (bact[it]) {
                const uint4 v = *(const uint4*)(buf + PA * APLANE + (size_t)(wave * UPWB + it * 64 + lane) * 16);
                bsum[0] += bf_lo(v.x); bsum[1] += bf_hi(v.x); bsum[2] += bf_lo(v.y); bsum[3] += bf_hi(v.y);
                bsum[4] += bf_lo(v.z); bsum[5] += bf_hi(v.z); bsum[6] += bf_lo(v.w); bsum[7] += bf_hi(v.w);
            }
        }
    };

    // ---- the walk: steps n = 0 .. len + 1 (input planes zs - 1 .. ze); step n's planes were requested PF steps earlier ----
    const std::integral_constant<int, 0> c0;
    const std::integral_constant<int, 1> c1;
    const std::integral_constant<int, 2> c2;
#pragma unroll
    for (int k = 0; k < PF; ++k) request(k);
    auto step = [&](auto ph, int n) {
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(WAITN) : "memory");   // plane n has landed (every wave's pieces); everyone is done with step n - 1
        request(n + PF);                                                              // into buffer (n - 1) % NB
        const char* buf = smem + (n % NB) * BUF;
        if (do_bias && n < len) bias_add(buf);
        compute(ph, buf, n < len);
    };
    for (int n = 0; n <= len + 1; n += 3) {
        step(c0, n);
        if (n + 1 > len + 1) break;
        step(c1, n + 1);
        if (n + 2 > len + 1) break;
        step(c2, n + 2);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");     // the last requests (zero pages) have landed: LDS is free for the epilogue

    // ---- sum the K-split waves of each pair through LDS, then the slab (as k_mfma_wgrad_z) ----
    constexpr int LDS_BYTES = NB * BUF;
    constexpr int TCH = (P * 27 * 1024 <= LDS_BYTES) ? 27 : ((P * 9 * 1024 <= LDS_BYTES) ? 9 : 3);
    static_assert(P * TCH * 1024 <= LDS_BYTES, "reduction scratch");
    float* red = (float*)smem;   // [P][TCH][64][4]
    if constexpr (WK > 1) {
#pragma unroll 1
        for (int kk = 1; kk < WK; ++kk) {
#pragma unroll
            for (int t0 = 0; t0 < T; t0 += TCH) {
                if (kw == kk) {
#pragma unroll
                    for (int t = 0; t < TCH; ++t) *(f32x4*)(red + ((pw * TCH + t) * 64 + lane) * 4) = acc[t0 + t];
                }
                __syncthreads();
                if (kw == 0) {
#pragma unroll
                    for (int t = 0; t < TCH; ++t) {
                        const f32x4 o = *(const f32x4*)(red + ((pw * TCH + t) * 64 + lane) * 4);
                        acc[t0 + t][0] += o[0]; acc[t0 + t][1] += o[1]; acc[t0 + t][2] += o[2]; acc[t0 + t][3] += o[3];
                    }
                }
                __syncthreads();
            }
        }
    }
    {
        constexpr int RP = 436;
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
            float* base = a.slab + (size_t)blockIdx.x * T * g.Cin * g.Cout + ((size_t)(cbB + pr % PB) * 16 * g.Cin + (size_t)(caB + pr / PB) * 16) * T;
            for (int q = tid; q < 16 * 108; q += 256) {
                const int row = q / 108, c4 = q % 108;
                *(f32x4*)(base + (size_t)row * g.Cin * T + c4 * 4) = *(const f32x4*)(stg + row * RP + c4 * 4);
            }
        }
    }
    if (do_bias) {
        // a thread's units are all of (tile, half) = (its wave's first unit / (NVB * 2), lane & 1): lanes of one parity, then the waves that share a tile
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = bsum[e];
#pragma unroll
            for (int m = 2; m < 64; m <<= 1) v += __shfl_xor(v, m);
            bsum[e] = v;
        }
        __syncthreads();
        float* bred = (float*)smem;   // [4 waves][2 halves][8]
        if (lane < 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) bred[(wave * 2 + lane) * 8 + e] = bsum[e];
        }
        __syncthreads();
        if (tid < PB * 16) {
            const int tile = tid / 16, half = (tid / 8) & 1, e = tid % 8;
            float sacc = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w)
                if ((w * UPWB) / (NVB * 2) == tile) sacc += bred[(w * 2 + half) * 8 + e];
            a.bias_slab[(size_t)blockIdx.x * g.Cout + (cbB + tile) * 16 + half * 8 + e] = sacc;
        }
    }
}

template <int WK, int PA, int PB>
static void launch_zd_t(const WgradZdArgs& a, int gx, int gy, hipStream_t s, int polite) {
    constexpr int BY = 2 * WK, lds = 4 * (PA * (BY + 2) * 34 * 32 + PB * BY * 32 * 32);
    static_assert(lds <= 80 * 1024, "two blocks per CU");
    static std::atomic<uint64_t> attr_done{0};
    set_max_lds_once(attr_done, (const void*)k_mfma_wgrad_zd<WK, PA, PB>, polite_lds(lds, 1));
    k_mfma_wgrad_zd<WK, PA, PB><<<dim3((unsigned)gx, (unsigned)gy), 256, polite_lds(lds, polite), s>>>(a);
}
// the 4-wave configurations of k_mfma_wgrad_z's work split (wk x pa x pb = 2x2x1, 2x1x2, 4x1x1) with LDS-DMA staging; false: not one of them
bool launch_wgrad_zd(const ConvGeom& g, const SrcDesc* src, int nsrc, const void* dy, float* slab, float* bias_slab, int wk, int pa, int pb, int cols_x,
                     int cols_y, int nseg, int zlen, int gx, int gy, hipStream_t s, int polite) {
    if (wk * pa * pb != 4) return false;
    for (int k = 0; k < nsrc; ++k) if ((size_t)g.H * g.W * src[k].C * 2 >= ((size_t)1 << 40)) return false;
    WgradZdArgs a;
    a.g = g; a.nasrc = nsrc; a.asrc[0] = src[0]; if (nsrc > 1) a.asrc[1] = src[1];
    a.dy = dy; a.slab = slab; a.bias_slab = bias_slab;
    a.cols_x = cols_x; a.cols_y = cols_y; a.nseg = nseg; a.zlen = zlen;
    if (wk == 2 && pa == 2) launch_zd_t<2, 2, 1>(a, gx, gy, s, polite);
    else if (wk == 2) launch_zd_t<2, 1, 2>(a, gx, gy, s, polite);
    else launch_zd_t<4, 1, 1>(a, gx, gy, s, polite);
    return true;
}

}  // namespace unet
