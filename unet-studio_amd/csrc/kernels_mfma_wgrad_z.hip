// Sliding-window weight gradient of the 3x3x3 stride-1 convolutions on the matrix cores (gfx950, bf16 storage, fp32 accumulate):
//
//     dW[co][ci][kz][ky][kx] = sum_v  a[v + (kz-1, ky-1, kx-1)][ci] * dy[v][co]          (autograd of unet.cpp:59-72, train.cpp:706)
//
// 27 implicit GEMMs  D_t[ca][cb] += A_t[ca][k] * B[k][cb]  with k = voxel; both tensors are channels-last, so both MFMA operands
// come from LDS planes [voxel][16 ch] through the transposing ds_read_b64_tr_b16 (as in k_mfma_wgrad, kernels_mfma_wgrad.hip).
//
// What is different from k_mfma_wgrad (halo tile per block, one A-fragment read pair PER MFMA: LDS-read bound at 1 KB per MFMA,
// 6-12 % of the MFMA peak at 128^3):
//   * a block owns a (BY x BX) footprint in (y, x) and WALKS z over a segment: one step stages ONE plane of the input (halo
//     (BY+2) x (BX+2)) and ONE plane of dy, double buffered, one barrier per plane -- 1.3x the minimum LDS / L2 traffic instead of 2.8x;
//   * a K-step is 32 voxels of a row (BX = 32) or of two rows (BX = 16).  The A fragment of (input plane p, row a, shift kx) is the
//     operand of tap (kz, ky) for the dy rows of plane p + 1 - kz, row a - ky: with the dy fragments of the three planes p-1, p,
//     p+1 held in REGISTERS (they rotate: each is read from LDS once), one A-fragment read pair feeds up to 9 MFMAs.  Per wave and
//     plane: 12 A + 2 B read pairs for 54 MFMAs (0.26 KB of LDS reads per MFMA);
//   * the 27 tap accumulators stay in registers for the whole segment; waves of a block split the rows (WK) and the
//     (ca-tile, cb-tile) pairs (PA x PB); K-split waves are summed through LDS at the end; one slab [cb][ca][t] per block, summed
//     in a fixed order by the reduce kernel (no float atomics: bit-reproducible).
// Planes outside the segment / volume enter as zeros (staged zeros for the input, zero fragments for dy), so every step runs the
// same instruction stream.
#include <cstdio>
#include <type_traits>

#include "mfma_util.h"
#ifndef WZ_RD
#define WZ_RD 4
#endif

namespace unet {

typedef __attribute__((ext_vector_type(4))) short zs16x4;
typedef __attribute__((address_space(3))) zs16x4 zlds_s16x4;

struct WgradZArgs {
    ConvGeom g;        // Cin = Ca (input channels), Cout = Cb (dy channels); D,H,W = volume (stride 1: input = output size)
    SrcDesc asrc[2];   // input (may be a channel concat), plain (activated copies)
    int nasrc;
    const void* dy;
    float* slab;       // [gridDim.x][Cb][Ca][27]
    float* bias_slab;  // [gridDim.x][Cb] or nullptr
    int cols_x, cols_y, nseg, zlen;
};

__device__ __forceinline__ bf16x8 ztr_read2(const char* p0, const char* p1) {
    zs16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((zlds_s16x4*)p0);
    zs16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((zlds_s16x4*)p1);
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int BX, int WK, int PA, int PB>
__global__ void __launch_bounds__(64 * WK * PA * PB, (WK * PA * PB) == 4 ? 2 : 2) k_mfma_wgrad_z(WgradZArgs a) {
    constexpr int NW = WK * PA * PB, NT = 64 * NW, P = PA * PB;
    constexpr int RPK = 32 / BX;                 // rows per K-step
    constexpr int R = 2;                         // K-steps per wave and plane
    constexpr int BY = R * WK * RPK, HY = BY + 2, HX = BX + 2;
    constexpr int AROW = HX * 32, APLANE = HY * AROW, BROW = BX * 32, BPLANE = BY * BROW;
    constexpr int BUF = PA * APLANE + PB * BPLANE;
    constexpr int T = 27;
    constexpr int NA = RPK * (R - 1) + 3;        // A fragments (first rows) a wave reads per plane and kx
    static_assert(BX == 32 || BX == 16, "row width");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, il = lane & 15, gq = lane >> 4;
    const int q4 = il >> 2, p4 = il & 3;
    const int pw = wave % P, kw = wave / P;
    const int ia = pw / PB, ib = pw % PB;
    const int CBG = (g.Cout / 16) / PB;                       // cb-tile groups
    const int caB = ((int)blockIdx.y / CBG) * PA, cbB = ((int)blockIdx.y % CBG) * PB;
    const int C0 = a.asrc[0].C;

    // this block's item: footprint column and z segment.  Blocks that share an XCD get a contiguous range of items, ordered
    // segment-major: a compact patch of columns of one z segment, whose shared (y, x) halos are hits in that XCD's L2
    const int item = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int ncols = a.cols_x * a.cols_y;
    const int seg = item / ncols, col = item % ncols;
    const int x0 = (col % a.cols_x) * BX, y0 = (col / a.cols_x) * BY;
    const int zs = seg * a.zlen, ze = zs + a.zlen < g.D ? zs + a.zlen : g.D, len = ze - zs;

    f32x4 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum[e] = 0.f;

    // ---- staging roles: a thread always stages the same (tile, 8-channel half) of A and of B ----
    constexpr int GA = PA * 2, GB = PB * 2;
    static_assert(NT % GA == 0 && NT % GB == 0, "fixed staging roles");
    const int ua = tid % GA, ub = tid % GB;
    const int ca = (caB + (ua >> 1)) * 16 + (ua & 1) * 8;
    const int sa = (a.nasrc > 1 && ca >= C0) ? 1 : 0;
    const int aC = sa ? a.asrc[1].C : C0;
    const char* abase = (const char*)(sa ? a.asrc[1].ptr : a.asrc[0].ptr) + (size_t)(ca - (sa ? C0 : 0)) * 2;
    const int cb = (cbB + (ub >> 1)) * 16 + (ub & 1) * 8;
    const char* bbase = (const char*)a.dy + (size_t)cb * 2;
    const bool do_bias = a.bias_slab != nullptr && caB == 0;
    const unsigned avs = (unsigned)aC * 2, bvs = (unsigned)g.Cout * 2;
    const size_t aplane_b = (size_t)g.H * g.W * avs, bplane_b = (size_t)g.H * g.W * bvs;

    constexpr int UNITS_A = PA * HY * HX * 2, ITERS_A = (UNITS_A + NT - 1) / NT;
    constexpr int UNITS_B = PB * BY * BX * 2, ITERS_B = (UNITS_B + NT - 1) / NT;
    int la[ITERS_A], lb[ITERS_B];          // LDS byte offset inside a buffer (-1: no unit)
    unsigned ga[ITERS_A], gb[ITERS_B];     // byte offset inside a plane of the source (valid units)
    unsigned amask = 0, bmask = 0;         // unit lies inside the volume in (y, x)
#pragma unroll
    for (int it = 0; it < ITERS_A; ++it) {
        const int u = tid + it * NT, hv = u / GA, hy = hv / HX, hx = hv % HX;
        const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        const bool in = u < UNITS_A;
        la[it] = in ? (ua >> 1) * APLANE + hv * 32 + (ua & 1) * 16 : -1;
        const bool ok = in && (unsigned)gy < (unsigned)g.H && (unsigned)gx < (unsigned)g.W;
        if (ok) amask |= 1u << it;
        ga[it] = ok ? (unsigned)(gy * g.W + gx) * avs : 0u;
    }
#pragma unroll
    for (int it = 0; it < ITERS_B; ++it) {
        const int u = tid + it * NT, tv = u / GB, ty = tv / BX, tx = tv % BX;
        const int gy = y0 + ty, gx = x0 + tx;
        const bool in = u < UNITS_B;
        lb[it] = in ? PA * APLANE + (ub >> 1) * BPLANE + tv * 32 + (ub & 1) * 16 : -1;
        const bool ok = in && gy < g.H && gx < g.W;
        if (ok) bmask |= 1u << it;
        gb[it] = ok ? (unsigned)(gy * g.W + gx) * bvs : 0u;
    }
    // Two register sets: the loads of plane n+2 are issued at the start of step n and stored to LDS at the end of step n+1, so two
    // planes per block are in flight (with one set -- a single step of flight -- the kernel ran at ~3.7 TB/s: 2 blocks x 19 KB per CU
    // do not cover HBM's latency under load).  The loads are inline assembly, waited for by hand (as in k_mfma_conv_z): left to the
    // compiler, the zero-select of the OLDER set was hoisted to the top of the step and its vmcnt(4..0) ladder drained the set that
    // had just been requested.  Rules that keep this safe (checked on the emitted code by tools/check_asm_loads.py at build time):
    //   * every step issues exactly NL loads per thread and nothing else that counts in vmcnt (no stores, no compiler loads): units
    //     outside the volume read a valid dummy address and are zeroed in LDS by a second store to the same address;
    //   * no VALU instruction touches a prefetch register: it goes from the load straight into ds_write_b128 (both inline
    //     assembly); the bias sums re-read the stored dy units from LDS (LDS operations of a wave execute in order).
    constexpr int NL = ITERS_A + ITERS_B;
    // PD register sets = planes in flight.  Three where the registers allow it (NL <= 4 loads per step: the multi-pair blocks, whose
    // steps are short -- 27 KB staged for 0.7 us of MFMAs -- and ran at ~1 TB/s with two); the one-pair block keeps two (5 loads per step).
    constexpr int PD = NL <= 4 ? 3 : 2;
    bf16x8 RA[PD][ITERS_A], RB[PD][ITERS_B];
    const bf16x8 zero8 = __builtin_bit_cast(bf16x8, make_uint4(0u, 0u, 0u, 0u));
    auto fetch = [&](auto setc, int pz, int bz) {
        constexpr int SET = decltype(setc)::value;
        auto& ra = RA[SET];    // named here: operands of an asm statement alone do not make a generic lambda capture the arrays
        auto& rb = RB[SET];
        const bool zin = (unsigned)pz < (unsigned)g.D;
        const char* ap = abase + (size_t)(zin ? pz : 0) * aplane_b;
#pragma unroll
        for (int it = 0; it < ITERS_A; ++it) {
            const char* ad = ap + ga[it];
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ra[it]) : "v"(ad) : "memory");
        }
        // dy planes past the segment are never stored: repeat the segment's last plane (an L2 hit) instead of reading two more from HBM
        const int bzc = bz < ze ? bz : ze - 1;
        const char* bp = bbase + (size_t)((unsigned)bzc < (unsigned)g.D ? bzc : 0) * bplane_b;
#pragma unroll
        for (int it = 0; it < ITERS_B; ++it) {
            const char* ad = bp + gb[it];
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rb[it]) : "v"(ad) : "memory");
        }
    };
    // stores set SET: input plane pz (zeros outside the volume), dy plane only when it lies inside the segment (have_b).
    // younger: how many fetches (NL loads each) were issued after this set's and may stay in flight.
    auto commit = [&](auto setc, char* buf, int pz, bool have_b, int younger) {
        constexpr int SET = decltype(setc)::value;
        auto& ra = RA[SET];
        auto& rb = RB[SET];
        const bool zin = (unsigned)pz < (unsigned)g.D;
        if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NL) : "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NL) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int it = 0; it < ITERS_A; ++it)
            if ((it + 1) * NT <= UNITS_A || la[it] >= 0) {
                char* d = buf + la[it];
                const unsigned dl = (unsigned)(size_t)(__attribute__((address_space(3))) char*)d;
                asm volatile("ds_write_b128 %0, %1" :: "v"(dl), "v"(ra[it]) : "memory");
                if (!(zin && ((amask >> it) & 1u))) *(bf16x8*)d = zero8;
            }
        if (have_b) {
#pragma unroll
            for (int it = 0; it < ITERS_B; ++it)
                if ((it + 1) * NT <= UNITS_B || lb[it] >= 0) {
                    char* d = buf + lb[it];
                    const unsigned dl = (unsigned)(size_t)(__attribute__((address_space(3))) char*)d;
                    asm volatile("ds_write_b128 %0, %1" :: "v"(dl), "v"(rb[it]) : "memory");
                    if (!((bmask >> it) & 1u)) *(bf16x8*)d = zero8;
                    if (do_bias) {
                        const uint4 v = *(const uint4*)d;
                        bsum[0] += bf_lo(v.x); bsum[1] += bf_hi(v.x); bsum[2] += bf_lo(v.y); bsum[3] += bf_hi(v.y);
                        bsum[4] += bf_lo(v.z); bsum[5] += bf_hi(v.z); bsum[6] += bf_lo(v.w); bsum[7] += bf_hi(v.w);
                    }
                }
        }
    };

    // ---- fragment addresses: lane group gq, read r fetch voxel group G = gq + 4r of the K-step (4 consecutive x) ----
    // row inside the K-step = G / (BX/4), x = (G % (BX/4)) * 4 + q4; lane 4q+p supplies row q (voxel q), columns 4p..4p+3
    int aoff[2], boff[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int G = gq + 4 * r, row = G / (BX / 4), x = (G % (BX / 4)) * 4 + q4;
        aoff[r] = ia * APLANE + ((kw * R * RPK + row) * HX + x) * 32 + p4 * 8;
        boff[r] = PA * APLANE + ib * BPLANE + ((kw * R * RPK + row) * BX + x) * 32 + p4 * 8;
    }

    bf16x8 Bq[3][R];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int j = 0; j < R; ++j) Bq[s][j] = zero8;

    // one input plane: PH = register slot of the NEW dy plane (kz = 0); kz = 1 -> slot PH+2, kz = 2 -> slot PH+1 (mod 3)
    auto compute = [&](auto ph, const char* buf, bool bnew) {
        constexpr int PH = decltype(ph)::value;
        const char* pa0 = buf + aoff[0];
        const char* pa1 = buf + aoff[1];
        const char* pb0 = buf + boff[0];
        const char* pb1 = buf + boff[1];
#pragma unroll
        for (int j = 0; j < R; ++j) Bq[PH][j] = bnew ? ztr_read2(pb0 + j * RPK * BROW, pb1 + j * RPK * BROW) : zero8;
        constexpr int NREAD = NA * 3, RD = WZ_RD;
        bf16x8 ring[RD];
#pragma unroll
        for (int i = 0; i < RD && i < NREAD; ++i) ring[i] = ztr_read2(pa0 + ((i / 3) * HX + i % 3) * 32, pa1 + ((i / 3) * HX + i % 3) * 32);
#pragma unroll
        for (int i = 0; i < NREAD; ++i) {
            const int ai = i / 3, kx = i % 3;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kz = 2; kz >= 0; --kz)                  // the new dy plane's fragments (kz = 0) are the last to be needed
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    const int ky = ai - RPK * j;
                    if (ky >= 0 && ky <= 2)
                        acc[kz * 9 + ky * 3 + kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[i % RD], Bq[(PH + (3 - kz)) % 3][j],
                                                                                            acc[kz * 9 + ky * 3 + kx], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
            if (i + RD < NREAD) {
                const int m = i + RD;
                ring[m % RD] = ztr_read2(pa0 + ((m / 3) * HX + m % 3) * 32, pa1 + ((m / 3) * HX + m % 3) * 32);
            }
        }
    };

    // ---- the walk: step n stages input plane zs-1+n+1 and dy plane zs+n+1 while it computes input plane zs-1+n ----
    // Step n computes input plane zs-1+n from buffer n&1 (with the dy planes zs+n-2 .. zs+n in registers), stores plane zs+n
    // (fetched during step n-1) to the other buffer and requests plane zs+n+1.  Six steps per loop trip as straight-line code:
    // 3 register slots of the new dy plane x 2 prefetch sets are all compile-time (with the slot chosen by a branch inside ONE
    // step body the three inlined bodies met in phi nodes of all 27 accumulators and the allocator spilled ~150 VGPRs).
    const std::integral_constant<int, 0> c0;
    const std::integral_constant<int, 1> c1;
    const std::integral_constant<int, 2> c2;
    // plane k (input zs-1+k, dy zs+k) travels in register set k % PD; step n stores plane n+1 and requests plane n+PD
    fetch(c0, zs - 1, zs);
    commit(c0, smem, zs - 1, len > 0, 0);
    fetch(c1, zs, zs + 1);
    if constexpr (PD == 3) fetch(c2, zs + 1, zs + 2);
    __syncthreads();
    auto step = [&](auto ph, auto set_commit, auto set_fetch, int n) {
        char* cur = smem + (n & 1) * BUF;
        char* nxt = smem + ((n + 1) & 1) * BUF;
        const bool more = n <= len;                          // another step follows: plane n+1 is needed
        const bool ahead = n + PD <= len + 1;                // plane n+PD is used by step n+PD
        if (ahead) fetch(set_fetch, zs - 1 + n + PD, zs + n + PD);
        compute(ph, cur, n < len);
        // fetches younger than plane n+1's: planes n+2 .. n+PD as far as they were requested (plane k exists for k <= len+1)
        int younger = len + 1 - (n + 1);
        if (younger > PD - 1) younger = PD - 1;
        if (younger < 0) younger = 0;
        if (more) commit(set_commit, nxt, zs + n, n + 1 < len, younger);
        __syncthreads();
    };
    if constexpr (PD == 2) {
        for (int n = 0; n <= len + 1; n += 6) {          // 3 dy slots x 2 sets
            step(c0, c1, c0, n);
            if (n + 1 > len + 1) break;
            step(c1, c0, c1, n + 1);
            if (n + 2 > len + 1) break;
            step(c2, c1, c0, n + 2);
            if (n + 3 > len + 1) break;
            step(c0, c0, c1, n + 3);
            if (n + 4 > len + 1) break;
            step(c1, c1, c0, n + 4);
            if (n + 5 > len + 1) break;
            step(c2, c0, c1, n + 5);
        }
    } else {
        for (int n = 0; n <= len + 1; n += 3) {          // 3 dy slots = 3 sets: plane n+1 sits in set (n+1) % 3, plane n+3 goes to set n % 3
            step(c0, c1, c0, n);
            if (n + 1 > len + 1) break;
            step(c1, c2, c1, n + 1);
            if (n + 2 > len + 1) break;
            step(c2, c0, c2, n + 2);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing may be in flight when the epilogue reuses the prefetch registers

    // ---- sum the K-split waves of each pair through LDS (taps in chunks that fit), then the slab ----
    constexpr int LDS_BYTES = 2 * BUF;
    constexpr int TCH = (P * 27 * 1024 <= LDS_BYTES) ? 27 : ((P * 9 * 1024 <= LDS_BYTES) ? 9 : 3);
    static_assert(P * TCH * 1024 <= LDS_BYTES, "reduction scratch");
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
    // Slab layout = the gradient's own layout [cb][ca][t]: for one (ca, cb) tile pair a cb row is 16*27 contiguous floats (1728 B).
    // A lane owns cb = il and ca = gq*4 .. +3, i.e. 16-B pieces 432 B apart -- stored straight from the registers, every wave
    // instruction touched 64 different lines (27 KB per pair in 1728 scattered pieces: ~10 us of a 2x2-pair block's life).  The
    // pair's tile is therefore transposed through LDS (row pitch 436 floats: 2-way conflicts at most) and leaves as whole rows.
    {
        constexpr int RP = 436;
        static_assert(16 * RP * 4 <= LDS_BYTES, "slab staging");
        float* stg = (float*)smem;
        const bool store = a.nseg >= 0;
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
            float* base = a.slab + (size_t)blockIdx.x * T * g.Cin * g.Cout +
                          ((size_t)(cbB + pr % PB) * 16 * g.Cin + (size_t)(caB + pr / PB) * 16) * T;
            if (store)
                for (int q = tid; q < 16 * 108; q += NT) {
                    const int row = q / 108, c4 = q % 108;
                    *(f32x4*)(base + (size_t)row * g.Cin * T + c4 * 4) = *(const f32x4*)(stg + row * RP + c4 * 4);
                }
        }
    }
    if (do_bias) {
        // threads with equal tid % GB hold partial sums of the same 8 channels: shuffle tree inside each wave (lanes GB apart), then
        // the NW wave totals through LDS.  (A serial loop of NT / GB LDS reads per output here cost ~6 us at the end of every block.)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = bsum[e];
#pragma unroll
            for (int m = GB; m < 64; m <<= 1) v += __shfl_xor(v, m);
            bsum[e] = v;
        }
        __syncthreads();
        float* bred = (float*)smem;   // [NW][GB][8]
        if (lane < GB) {
#pragma unroll
            for (int e = 0; e < 8; ++e) bred[(wave * GB + lane) * 8 + e] = bsum[e];
        }
        __syncthreads();
        if (tid < GB * 8) {
            const int u = tid / 8, e = tid % 8;
            float sacc = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) sacc += bred[(w * GB + u) * 8 + e];
            a.bias_slab[(size_t)blockIdx.x * g.Cout + (cbB + (u >> 1)) * 16 + (u & 1) * 8 + e] = sacc;
        }
    }
}

// ---- configuration / launch ----
struct WgradZCfg { int bx, wk, pa, pb, by, cols_x, cols_y, nseg, zlen, gx, gy; };

static bool wgrad_z_cfg(const ConvGeom& g, WgradZCfg& c, int polite = 0) {
    if (sliding_window_off() || g.ks != 3 || g.stride != 1 || g.Cin % 16 || g.Cout % 16) return false;
    if (g.W < 24 || g.D < 4) return false;                 // narrower volumes: k_mfma_wgrad
    const int cat = g.Cin / 16, cbt = g.Cout / 16;
    c.bx = 32;
    if (cat % 2 == 0 && cbt % 2 == 0) { c.pa = 2; c.pb = 2; c.wk = 2; }
    else if (cat % 2 == 0) { c.pa = 2; c.pb = 1; c.wk = 4; }
    else if (cbt % 2 == 0) { c.pa = 1; c.pb = 2; c.wk = 4; }
    else { c.pa = 1; c.pb = 1; c.wk = 4; }
    // at 32^3 a 2x2-pair block leaves 128 blocks for the chip (4 footprint columns x 8 segments x 4 pair groups); single pairs give 512
    // (step 3.22 -> 3.19 ms; at 64^3 no difference)
    if ((int64_t)g.D * g.H * g.W <= (int64_t)32768) { c.pa = 1; c.pb = 1; c.wk = 4; }
    else if (polite) {   // 4-wave blocks: a pair of tiles on one side where the channels allow it (one read of the other side serves both), 4 rows
        if (cat % 2 == 0) { c.pa = 2; c.pb = 1; c.wk = 2; }
        else if (cbt % 2 == 0) { c.pa = 1; c.pb = 2; c.wk = 2; }
        else { c.pa = 1; c.pb = 1; c.wk = 4; }
    }
    c.by = 2 * c.wk;
    c.cols_x = (g.W + c.bx - 1) / c.bx; c.cols_y = (g.H + c.by - 1) / c.by;
    c.gy = (cat / c.pa) * (cbt / c.pb);
    const int cols = c.cols_x * c.cols_y;
    const int nwaves = c.wk * c.pa * c.pb;
    // ~8 waves per CU in total; a polite launch is resident in one round at one block per CU
    int want = nwaves == 4 ? (polite ? 256 : 512) : 256;
    want /= c.gy;
    if (want < 1) want = 1;
    int nseg = (want + cols - 1) / cols;
    if (nseg < 1) nseg = 1;
    int zlen = (g.D + nseg - 1) / nseg;
    if (zlen < 4) zlen = 4;                                // >= 4 planes of work per 2 warm-up steps
    if (zlen > g.D) zlen = g.D;
    c.nseg = (g.D + zlen - 1) / zlen; c.zlen = zlen;
    c.gx = cols * c.nseg;
    return true;
}
bool mfma_wgrad_z_supported(int dtype, const ConvGeom& g, const SrcDesc* src, int nsrc) {
    WgradZCfg c;
    if (dtype != 1 || !wgrad_z_cfg(g, c)) return false;
    for (int s = 0; s < nsrc; ++s)
        if (src[s].C % 16 || src[s].scale || src[s].act) return false;
    return true;
}
int mfma_wgrad_z_splits(const ConvGeom& g, int polite) {
    WgradZCfg c;
    return wgrad_z_cfg(g, c, polite) ? c.gx : 0;
}
size_t mfma_wgrad_z_scratch_bytes(const ConvGeom& g, int polite) {
    WgradZCfg c;
    if (!wgrad_z_cfg(g, c, polite)) return 0;
    return ((size_t)c.gx * 27 * g.Cin * g.Cout + (size_t)c.gx * g.Cout) * 4 + 256;
}

template <int BX, int WK, int PA, int PB>
static void launch_wz(const WgradZArgs& a, const WgradZCfg& c, hipStream_t s, int polite) {
    constexpr int RPK = 32 / BX, BY = 2 * WK * RPK, HY = BY + 2, HX = BX + 2;
    constexpr int lds = 2 * (PA * HY * HX * 32 + PB * BY * BX * 32);
    static_assert(lds <= 80 * 1024, "LDS budget");
    static_assert(WK * PA * PB == 8, "the 4-wave work splits run on k_mfma_wgrad_zd");
    static std::atomic<uint64_t> attr_done{0};
    set_max_lds_once(attr_done, (const void*)k_mfma_wgrad_z<BX, WK, PA, PB>, lds);
    (void)polite;      // 8-wave blocks are never launched politely (wgrad_z_cfg gives polite launches a 4-wave split)
    k_mfma_wgrad_z<BX, WK, PA, PB><<<dim3((unsigned)c.gx, (unsigned)c.gy), 64 * WK * PA * PB, lds, s>>>(a);
}

// Launches the kernel only: slab [gx][Cout][Cin][27] (+ bias_slab [gx][Cout] when want_bias) at `scratch`; returns the number of
// slab rows (0: shape not covered).  The caller sums the rows (wgrad_reduce / the plan's batched reduce).
int launch_mfma_wgrad_z(const ConvGeom& g, const SrcDesc* src, int nsrc, const void* dy, bool want_bias, void* scratch, hipStream_t s,
                        int polite) {
    WgradZCfg c;
    if (!wgrad_z_cfg(g, c, polite)) return 0;
    WgradZArgs a;
    a.g = g; a.nasrc = nsrc; a.asrc[0] = src[0]; if (nsrc > 1) a.asrc[1] = src[1];
    a.dy = dy;
    a.slab = (float*)scratch;
    a.bias_slab = want_bias ? a.slab + (size_t)c.gx * 27 * g.Cin * g.Cout : nullptr;
    a.cols_x = c.cols_x; a.cols_y = c.cols_y; a.nseg = c.nseg; a.zlen = c.zlen;
    // the 4-wave work splits (polite launches, <= 32^3 volumes, odd tile counts) run with LDS-DMA staging: kernels_mfma_wgrad_zd.hip
    if (launch_wgrad_zd(g, src, nsrc, dy, a.slab, a.bias_slab, c.wk, c.pa, c.pb, c.cols_x, c.cols_y, c.nseg, c.zlen, c.gx, c.gy, s, polite)) return c.gx;
    if (c.pa == 2 && c.pb == 2) launch_wz<32, 2, 2, 2>(a, c, s, polite);
    else if (c.pa == 2) launch_wz<32, 4, 2, 1>(a, c, s, polite);
    else launch_wz<32, 4, 1, 2>(a, c, s, polite);
    return c.gx;
}

}  // namespace unet
