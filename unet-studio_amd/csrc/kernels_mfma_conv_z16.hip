// Sliding-window 3x3x3 stride-1 convolution for a SINGLE 16-channel chunk (Cin = 16): the 16->16 layers at full resolution, forward
// and dgrad, and the 16->32 dgrad of decode0.0.  k_mfma_conv_p serves these with 6x10x18 halo tiles (2.1x the outputs staged
// through LDS) and one 1-KB LDS read per MFMA: at Cout = 16 a patch fragment feeds a single MFMA, and the kernel is LDS-read bound
// (49 us for 134 MB at 128^3).
//
// Input-stationary along z: a block owns an 8x16 (y, x) footprint and walks z.  When input plane q is resident, its fragments are
// read ONCE -- five k-steps (the nine (ky, kx) taps in pairs: K = 32 = two taps x 16 channels) per m-tile -- and every fragment
// feeds three MFMAs, one per kz, into the accumulators of the three output planes q+1, q, q-1 that plane q contributes to.  The
// accumulators rotate through registers (3 planes x 2 m-tiles x 4 VGPRs); plane q-1 is complete after step q and leaves through
// the epilogue.  LDS reads per MFMA: 1/3 KB instead of 1 KB.
//
// Planes travel HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, five planes ahead, a ring of six 5.6-KB planes): no staging registers
// and no ds_write pass; the DMA is inline assembly waited for with a hand-counted vmcnt (see `dma` / `step`).  Blocks of an XCD own
// a compact patch of columns of one z segment, so the halos they share are L2 hits.
//
// The filter is the ordinary CK = 16 pack ([kstep][row tile][lane][8], taps paired in sequence 2ks, 2ks+1): the pairs this kernel
// needs -- inside one kz -- are gathered lane by lane when the fragments are loaded, once per block.
// Same arguments, epilogue semantics (bias, bf16, two destinations, accumulate, statistics rows per blockIdx.x) as k_mfma_conv_p.
#include <cstdlib>
#include <type_traits>

#include "mfma_util.h"

namespace unet {

// what LDS-DMA lanes outside the volume read (16 B of zeros, L2-resident)
__device__ __attribute__((aligned(16))) unsigned g_z16_zero[4] = {0u, 0u, 0u, 0u};

template <int NT, bool BNS>
__global__ void __launch_bounds__(256, 2) k_mfma_conv_z16(MfmaConvArgs a, ZWork zw) {
    constexpr int BY = 8, BX = 16, HY = BY + 2, HX = BX + 2, PLANE_B = HY * HX * 32;
    // a plane = 360 16-B units, LDS image linear in unit order; wave w moves units [90 w, 90 w + 90): one 64-lane LDS-DMA and one of 26 lanes
    constexpr int UNITS = HY * HX * 2, UPW = UNITS / 4, ITERS = 2;
    constexpr int NBUF = 6, PF = 5;                 // ring of planes in LDS, planes requested ahead (NBUF >= PF + 1)
    constexpr int UOFF = NBUF * PLANE_B, UTILE_B = BY * BX * 32;      // BNS: ring of raw-tensor tiles (one 64-lane DMA piece per wave)
    static_assert(UTILE_B == 4 * 64 * 16, "one piece per wave");
    static_assert(UPW == 90 && NBUF % 3 == 0 && NBUF >= PF + 1, "unit split / ring");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, gq = lane >> 4, h = gq >> 1, lg = lane & 1;
    const int nt0 = blockIdx.y * NT, NTT = g.Cout / 16, C0 = a.src[0].C;
    const bf16x8* wp = (const bf16x8*)a.w;
    const bf16x8 zero8 = __builtin_bit_cast(bf16x8, make_uint4(0u, 0u, 0u, 0u));
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;       // LDS byte address of the ring

    // filter fragments wf[kz][s][n]: lanes of k-groups 0,1 hold tap 9 kz + 2s, k-groups 2,3 tap 9 kz + 2s + 1 (none for s = 4: zeros)
    bf16x8 wf[3][5][NT];
#pragma unroll
    for (int kz = 0; kz < 3; ++kz)
#pragma unroll
        for (int s = 0; s < 5; ++s)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int t = 2 * s + h, tg = kz * 9 + (t <= 8 ? t : 8);
                const bf16x8 v = wp[((size_t)(tg >> 1) * NTT + nt0 + n) * 64 + ((lane & 31) | ((tg & 1) << 5))];
                wf[kz][s][n] = t <= 8 ? v : zero8;
            }
    // patch address of k-step s in a plane for this lane's first m-tile (row 2 wave); the second m-tile is one row further
    int mb[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int t = 2 * s + h <= 8 ? 2 * s + h : 8, ky = t / 3, kx = t % 3;    // (s = 4, upper k-groups: any resident data, times zero)
        mb[s] = ((2 * wave + ky) * HX + j + kx) * 32 + (gq & 1) * 16;
    }
    // staging units of a plane (this lane's two units) and the wave-uniform LDS offsets of the two DMA pieces
    int uyx[ITERS];
    bool uact[ITERS];
    unsigned upiece[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int u = wave * UPW + it * 64 + lane, hv = u >> 1, hy = hv / HX, hx = hv % HX;
        uact[it] = it * 64 + lane < UPW;
        uyx[it] = hy | (hx << 8);
        upiece[it] = (unsigned)__builtin_amdgcn_readfirstlane((wave * UPW + it * 64) * 16);
    }
    const int c = lg * 8, sidx = (a.nsrc > 1 && c >= C0) ? 1 : 0;
    const char* sptr = (const char*)(sidx ? a.src[1].ptr : a.src[0].ptr) + (size_t)(c - (sidx ? C0 : 0)) * 2;
    const unsigned vstride = (unsigned)(sidx ? a.src[1].C : C0) * 2;

    float s1[NT][4], s2[NT][4], b4[NT][4];
    __amdgpu_buffer_rsrc_t orsrc[NT];
    int oC[NT], cd[NT], oacc[NT];
    bool ohave[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int cch = (nt0 + n) * 16 + gq * 4;                  // this lane's 4 output channels of row tile n
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[n][r] = 0.f; s2[n][r] = 0.f; b4[n][r] = a.bias ? a.bias[cch + r] : 0.f; }
        const int dsel = (a.nout > 1 && (nt0 + n) * 16 >= a.outC[0]) ? 1 : 0;      // uniform: destinations split at a multiple of 16
        char* ob = (char*)(dsel ? a.out[1] : a.out[0]);
        oC[n] = dsel ? a.outC[1] : a.outC[0];
        oacc[n] = dsel ? a.out_acc[1] : a.out_acc[0];
        cd[n] = cch - (dsel ? a.outC[0] : 0);
        ohave[n] = ob != nullptr;
        // outputs leave through a buffer descriptor: a lane outside the volume (or a plane outside the segment) stores beyond
        // num_records and the hardware drops it -- no branch in the step
        orsrc[n] = __builtin_amdgcn_make_buffer_rsrc(ob, 0, ob ? (int)((size_t)a.oD * a.oH * a.oW * oC[n] * 2) : 0, 0x00020000);
    }
    constexpr int OOB = (int)0x80000000;
    // BNS: this lane's four channels of the norm being differentiated, and its unit of a raw-tensor tile (row, x, 8-channel half)
    float bmean[4], brstd[4], bsc[4], bsh[4];
    if constexpr (BNS) {
        static_assert(NT == 1, "one row tile per block");
        const int cc = nt0 * 16 + gq * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            bmean[r] = a.bn_stat[cc + r]; brstd[r] = a.bn_stat[a.bn_C + cc + r];
            bsc[r] = a.bn_stat[2 * a.bn_C + cc + r]; bsh[r] = a.bn_stat[3 * a.bn_C + cc + r];
        }
    }
    const int bu_row = tid >> 5, bu_x = (tid >> 1) & 15;
    const unsigned bu_piece = (unsigned)__builtin_amdgcn_readfirstlane(wave * 64 * 16);

    const int nitems = zw.cols_x * zw.cols_y * zw.nseg;
    // Blocks of one XCD (blockIdx.x % 8) take a contiguous range of items, and items are ordered segment-major: an XCD then owns a
    // compact patch of columns of ONE z segment, marching z together, and the (y, x) halos its blocks share are L2 hits.  Dealt
    // round-robin, every halo was fetched by a different XCD: PMC 101 MB read per 67 MB input at 128^3 (1.41 x 1.06 = the halo
    // ratio); with the patches 72 MB.
    const int ncols = zw.cols_x * zw.cols_y;
    for (int item = xcd_remap(blockIdx.x, gridDim.x); item < nitems; item += gridDim.x) {
        const int seg = item / ncols, col = item % ncols;
        const int x0 = (col % zw.cols_x) * BX, y0 = (col / zw.cols_x) * BY;
        const int zs = seg * zw.zlen, ze = zs + zw.zlen < g.D ? zs + zw.zlen : g.D;      // output planes [zs, ze)
        bool uok[ITERS];
        const char* ubase[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int gy = y0 - 1 + (uyx[it] & 255), gx = x0 - 1 + (uyx[it] >> 8);
            uok[it] = uact[it] && (unsigned)gy < (unsigned)g.H && (unsigned)gx < (unsigned)g.W;
            ubase[it] = sptr + (size_t)(uok[it] ? gy * g.W + gx : 0) * vstride;
        }
        const size_t plane_bytes = (size_t)g.H * g.W * vstride;
        const int rlast = (ze - zs) + 2;          // last computing step; planes 0 .. rlast - 1 (zs - 1 .. ze) are needed
        // Plane r (relative to zs - 1) travels HBM -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass),
        // PF planes ahead of its use.  Units outside the volume, planes outside it and planes past the segment read a zero page.
        // The DMA is inline assembly, so hipcc neither counts it nor waits for it: every step issues exactly 2 DMA pieces and 2 output
        // stores per wave (stores outside the segment go to an out-of-range buffer offset), and plane r is awaited with
        // s_waitcnt vmcnt(4 PF - 2) = everything younger than its second piece.
        auto dma = [&](int r, int slot) {
            const int pz = zs - 1 + r;
            const bool zin = (unsigned)pz < (unsigned)g.D && r < rlast;
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const char* src = (zin && uok[it]) ? ubase[it] + (size_t)pz * plane_bytes : (const char*)g_z16_zero;
                const unsigned dst = lds0 + (unsigned)slot * PLANE_B + upiece[it];
                unsigned keep;
                if (it == 0 || uact[it])
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
            }
            if constexpr (BNS) {      // the raw-tensor tile of the output plane that completes when plane r is computed on: zs + r - 2
                const int k = r - 2, oy = y0 + bu_row, oxx = x0 + bu_x;
                const bool ok = k >= 0 && zs + k < ze && oy < a.oH && oxx < a.oW;
                const char* src = ok ? (const char*)a.bn_u + ((((size_t)(zs + k) * a.oH + oy) * a.oW + oxx) * a.bn_C + nt0 * 16 + lg * 8) * 2
                                     : (const char*)g_z16_zero;
                const unsigned dst = lds0 + UOFF + (unsigned)slot * UTILE_B + bu_piece;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
            }
        };
        // output offsets of this lane's two voxels (rows 2 wave, 2 wave + 1) in plane zs; plane o = zs + k adds k planes
        const int ox = x0 + j;
        bool ook[2];
        unsigned ooff[2][NT], oplane[NT];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int oy = y0 + 2 * wave + i;
            ook[i] = oy < a.oH && ox < a.oW;
#pragma unroll
            for (int n = 0; n < NT; ++n)
                ooff[i][n] = (unsigned)(((((size_t)zs * a.oH + (ook[i] ? oy : 0)) * a.oW + (ook[i] ? ox : 0)) * oC[n] + cd[n]) * 2);
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) oplane[n] = (unsigned)((size_t)a.oH * a.oW * oC[n] * 2);

        f32x4 acc[3][2][NT];
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[q][i][n] = f32x4{0.f, 0.f, 0.f, 0.f};

        // step r (phase PH = r mod 3): the fragments of plane r - 1 (ring slot BQ) feed output planes r (kz = 0, a fresh
        // accumulator), r - 1 and r - 2 (relative to zs - 1: output plane zs + k is k + 1); the latter is complete and leaves
        auto compute = [&](int r, auto phc, auto slc) {
            constexpr int PH = decltype(phc)::value, BQ = decltype(slc)::value;
            bf16x8 xr[2][5];
#pragma unroll
            for (int s = 0; s < 5; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i) xr[i][s] = *(const bf16x8*)(smem + BQ * PLANE_B + mb[s] + i * HX * 32);
            // all ten reads are in flight before the first MFMA (left alone the scheduler issues them two at a time, each pair followed by
            // a wait: five exposed LDS latencies per step at two waves per SIMD)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 5; ++s)
#pragma unroll
                for (int kz = 0; kz < 3; ++kz)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            f32x4& d = acc[(PH - kz + 3) % 3][i][n];
                            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kz][s][n], xr[i][s], (kz == 0 && s == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : d, 0, 0, 0);
                        }
            // epilogue of output plane k = r - 3 (zs + k), accumulator (PH + 1) % 3
            const int k = r - 3;
            const bool kin = k >= 0 && zs + k < ze;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const f32x4 d = acc[(PH + 1) % 3][i][n];
                    const bool ok = kin && ook[i] && ohave[n];
                    const int off = ok ? (int)(ooff[i][n] + (unsigned)k * oplane[n]) : OOB;
                    float v0 = d[0] + b4[n][0], v1 = d[1] + b4[n][1], v2 = d[2] + b4[n][2], v3 = d[3] + b4[n][3];
                    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
                    if (oacc[n]) {     // uniform; an out-of-range offset reads zeros
                        const u32x2 old = __builtin_amdgcn_raw_buffer_load_b64(orsrc[n], off, 0, 0);
                        v0 += bf_lo(old.x); v1 += bf_hi(old.x); v2 += bf_lo(old.y); v3 += bf_hi(old.y);
                    }
                    u32x2 o;
                    o.x = pack_bf16x2(v0, v1); o.y = pack_bf16x2(v2, v3);
                    __builtin_amdgcn_raw_buffer_store_b64(o, orsrc[n], off, 0, 0);
                    const float r0 = ok ? bf_lo(o.x) : 0.f, r1 = ok ? bf_hi(o.x) : 0.f, r2 = ok ? bf_lo(o.y) : 0.f, r3 = ok ? bf_hi(o.y) : 0.f;
                    if constexpr (BNS) {
                        // k_norm_bwd_stats8's sums from the gradient as it is stored (bf16) and the raw tensor's tile (zeros outside the volume)
                        const uint2 uu = *(const uint2*)(smem + UOFF + BQ * UTILE_B + ((2 * wave + i) * BX + j) * 32 + gq * 8);
                        const float uf[4] = {bf_lo(uu.x), bf_hi(uu.x), bf_lo(uu.y), bf_hi(uu.y)}, rr[4] = {r0, r1, r2, r3};
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float dv = rr[c] * act_d(fmaf(uf[c], bsc[c], bsh[c]), a.bn_act);
                            s1[n][c] += dv;
                            s2[n][c] = fmaf(dv, (uf[c] - bmean[c]) * brstd[c], s2[n][c]);
                        }
                    } else {
                        s1[n][0] += r0; s1[n][1] += r1; s1[n][2] += r2; s1[n][3] += r3;
                        s2[n][0] = fmaf(r0, r0, s2[n][0]); s2[n][1] = fmaf(r1, r1, s2[n][1]);
                        s2[n][2] = fmaf(r2, r2, s2[n][2]); s2[n][3] = fmaf(r3, r3, s2[n][3]);
                    }
                }
        };
        // step r: wait for plane r - 1, barrier (every wave's pieces have landed; every wave is done with plane r - 2, whose slot
        // plane r - 1 + PF overwrites: NBUF >= PF + 1), request plane r - 1 + PF, compute on plane r - 1.
        auto step = [&](int r, auto phc, auto slc) {
            constexpr int SL = decltype(slc)::value;      // slot of plane r - 1
            static_assert(PF == 5, "vmcnt below = (P + 2) PF - P, P = 2 pieces (3 with the raw-tensor tile)");
            if constexpr (BNS) asm volatile("s_waitcnt vmcnt(22)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(18)\n\ts_barrier" ::: "memory");
            dma(r - 1 + PF, (SL + PF) % NBUF);
            compute(r, phc, slc);
        };
        // the previous item's planes are no longer read, and its last requests (never used) have landed: slots can be refilled
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int r = 0; r < PF; ++r) {
            dma(r, r);
#pragma unroll
            for (int i = 0; i < 2 * NT; ++i) {      // two out-of-range stores: the prologue's operations count like a step's
                typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{0u, 0u}, orsrc[0], OOB + 8 * (r * 2 * NT + i), 0, 0);   // (distinct: identical stores are merged)
            }
        }
        // unrolled by 6 = lcm(3 accumulator phases, NBUF ring slots); plane r - 1 sits in slot (r - 1) % 6
        for (int r = 1; r <= rlast; r += 6) {
            step(r, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
            if (r + 1 > rlast) break;
            step(r + 1, std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{});
            if (r + 2 > rlast) break;
            step(r + 2, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
            if (r + 3 > rlast) break;
            step(r + 3, std::integral_constant<int, 1>{}, std::integral_constant<int, 3>{});
            if (r + 4 > rlast) break;
            step(r + 4, std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{});
            if (r + 5 > rlast) break;
            step(r + 5, std::integral_constant<int, 0>{}, std::integral_constant<int, 5>{});
        }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");    // outstanding requests write LDS: let them land before it is reused
    float* const srows = BNS ? a.bn_partial : a.stats;
    const int srowC = BNS ? a.bn_C : g.Cout;
    if (srows) {
        float* red = (float*)smem;
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float u = s1[n][r], v = s2[n][r];
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) { u += __shfl_xor(u, m); v += __shfl_xor(v, m); }
                if (j == 0) { red[((wave * NT + n) * 16 + gq * 4 + r) * 2] = u; red[((wave * NT + n) * 16 + gq * 4 + r) * 2 + 1] = v; }
            }
        __syncthreads();
        if (tid < NT * 16) {
            float u = 0.f, v = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { u += red[((w * NT) * 16 + tid) * 2]; v += red[((w * NT) * 16 + tid) * 2 + 1]; }
            srows[((size_t)blockIdx.x * srowC + nt0 * 16 + tid) * 2 + 0] = u;
            srows[((size_t)blockIdx.x * srowC + nt0 * 16 + tid) * 2 + 1] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same pipeline for a single 32-channel chunk (Cin = 32: 32->16 at full resolution -- the forward's heaviest launch -- and the
// 32->32 layers one level down, forward and dgrad; one 16-row tile per block, blockIdx.y = row tile).  K = 32 is one tap: nine
// k-steps per kz, 27 filter fragments in registers.  A wave's two m-tiles are consecutive rows, so tap ky of row 2w+1 reads the
// patch of tap ky+1 of row 2w: 12 fragment reads (4 rows x 3 kx) feed 54 MFMAs.  64-B voxels: the 16-B channel groups of a voxel
// are stored XOR-ed with bit 2 of the voxel's linear index (conflict-free ds_read_b128 for 16 consecutive voxels at any offset);
// the LDS image of an LDS-DMA is lane-linear, so the swizzle is applied to the SOURCE address (which channel group a lane fetches).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2) k_mfma_conv_z32(MfmaConvArgs a, ZWork zw) {
    constexpr int BY = 8, BX = 16, HY = BY + 2, HX = BX + 2, PLANE_B = HY * HX * 64;
    constexpr int UNITS = HY * HX * 4, UPW = UNITS / 4, ITERS = 3;     // 720 units; a wave moves 180: pieces of 64, 64 and 52 lanes
    constexpr int NBUF = 6, PF = 5;
    static_assert(UPW == 180 && UPW % 4 == 0 && NBUF % 3 == 0 && NBUF >= PF + 1, "unit split / ring");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, gq = lane >> 4;
    const int nt0 = blockIdx.y, NTT = g.Cout / 16, C0 = a.src[0].C;
    const bf16x8* wp = (const bf16x8*)a.w;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    bf16x8 wf[27];
#pragma unroll
    for (int ks = 0; ks < 27; ++ks) wf[ks] = wp[((size_t)ks * NTT + nt0) * 64 + lane];
    // patch addresses: plane rows 2 wave + R (R = 0..3), kx = 0..2
    int mb[4][3];
#pragma unroll
    for (int R = 0; R < 4; ++R)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int hv = (2 * wave + R) * HX + j + kx;
            mb[R][kx] = hv * 64 + ((gq ^ (((hv >> 2) & 1) << 1)) << 4);
        }
    // staging units: unit u = (voxel hv, slot); slot holds channel group slot ^ swz(hv)
    int uyx[ITERS];
    bool uact[ITERS];
    unsigned upiece[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int u = wave * UPW + it * 64 + lane, hv = u >> 2, hy = hv / HX, hx = hv % HX;
        uact[it] = it * 64 + lane < UPW;
        uyx[it] = hy | (hx << 8) | ((((u & 3) ^ (((hv >> 2) & 1) << 1))) << 16);       // hy, hx, channel group of the unit
        upiece[it] = (unsigned)__builtin_amdgcn_readfirstlane((wave * UPW + it * 64) * 16);
    }

    float s1[4], s2[4], b4[4];
    const int cch = nt0 * 16 + gq * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[r] = 0.f; s2[r] = 0.f; b4[r] = a.bias ? a.bias[cch + r] : 0.f; }
    const int dsel = (a.nout > 1 && nt0 * 16 >= a.outC[0]) ? 1 : 0;
    char* ob = (char*)(dsel ? a.out[1] : a.out[0]);
    const int oC = dsel ? a.outC[1] : a.outC[0], oacc = dsel ? a.out_acc[1] : a.out_acc[0], cd = cch - (dsel ? a.outC[0] : 0);
    const bool ohave = ob != nullptr;
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(ob, 0, ob ? (int)((size_t)a.oD * a.oH * a.oW * oC * 2) : 0, 0x00020000);
    constexpr int OOB = (int)0x80000000;

    const int nitems = zw.cols_x * zw.cols_y * zw.nseg, ncols = zw.cols_x * zw.cols_y;
    for (int item = xcd_remap(blockIdx.x, gridDim.x); item < nitems; item += gridDim.x) {
        const int seg = item / ncols, col = item % ncols;
        const int x0 = (col % zw.cols_x) * BX, y0 = (col / zw.cols_x) * BY;
        const int zs = seg * zw.zlen, ze = zs + zw.zlen < g.D ? zs + zw.zlen : g.D;
        bool uok[ITERS];
        const char* ubase[ITERS];
        unsigned uvs[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int gy = y0 - 1 + (uyx[it] & 255), gx = x0 - 1 + ((uyx[it] >> 8) & 255), c = (uyx[it] >> 16) * 8;
            const int sidx = (a.nsrc > 1 && c >= C0) ? 1 : 0;
            uok[it] = uact[it] && (unsigned)gy < (unsigned)g.H && (unsigned)gx < (unsigned)g.W;
            uvs[it] = (unsigned)(sidx ? a.src[1].C : C0) * 2;
            ubase[it] = (const char*)(sidx ? a.src[1].ptr : a.src[0].ptr) + (size_t)(c - (sidx ? C0 : 0)) * 2 + (size_t)(uok[it] ? gy * g.W + gx : 0) * uvs[it];
        }
        const size_t hw = (size_t)g.H * g.W;
        const int rlast = (ze - zs) + 2;
        // 3 DMA pieces + 2 output stores per wave and step: plane r is awaited with vmcnt(5 PF - 3) (see k_mfma_conv_z16)
        auto dma = [&](int r, int slot) {
            const int pz = zs - 1 + r;
            const bool zin = (unsigned)pz < (unsigned)g.D && r < rlast;
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const char* src = (zin && uok[it]) ? ubase[it] + (size_t)pz * hw * uvs[it] : (const char*)g_z16_zero;
                const unsigned dst = lds0 + (unsigned)slot * PLANE_B + upiece[it];
                unsigned keep;
                if (it < 2 || uact[it])
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
            }
        };
        const int ox = x0 + j;
        bool ook[2];
        unsigned ooff[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int oy = y0 + 2 * wave + i;
            ook[i] = oy < a.oH && ox < a.oW;
            ooff[i] = (unsigned)(((((size_t)zs * a.oH + (ook[i] ? oy : 0)) * a.oW + (ook[i] ? ox : 0)) * oC + cd) * 2);
        }
        const unsigned oplane = (unsigned)((size_t)a.oH * a.oW * oC * 2);
        f32x4 acc[3][2];
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[q][i] = f32x4{0.f, 0.f, 0.f, 0.f};

        auto compute = [&](int r, auto phc, auto slc) {
            constexpr int PH = decltype(phc)::value, BQ = decltype(slc)::value;
            bf16x8 xr[4][3];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int R = 0; R < 4; ++R) xr[R][kx] = *(const bf16x8*)(smem + BQ * PLANE_B + mb[R][kx]);
            __builtin_amdgcn_sched_barrier(0);       // all twelve reads in flight before the first MFMA
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kz = 0; kz < 3; ++kz)
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            f32x4& d = acc[(PH - kz + 3) % 3][i];
                            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kz * 9 + ky * 3 + kx], xr[i + ky][kx],
                                                                        (kz == 0 && kx == 0 && ky == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : d, 0, 0, 0);
                        }
            const int k = r - 3;
            const bool kin = k >= 0 && zs + k < ze;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f32x4 d = acc[(PH + 1) % 3][i];
                const bool ok = kin && ook[i] && ohave;
                const int off = ok ? (int)(ooff[i] + (unsigned)k * oplane) : OOB;
                float v0 = d[0] + b4[0], v1 = d[1] + b4[1], v2 = d[2] + b4[2], v3 = d[3] + b4[3];
                typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
                if (oacc) {
                    const u32x2 old = __builtin_amdgcn_raw_buffer_load_b64(orsrc, off, 0, 0);
                    v0 += bf_lo(old.x); v1 += bf_hi(old.x); v2 += bf_lo(old.y); v3 += bf_hi(old.y);
                }
                u32x2 o;
                o.x = pack_bf16x2(v0, v1); o.y = pack_bf16x2(v2, v3);
                __builtin_amdgcn_raw_buffer_store_b64(o, orsrc, off, 0, 0);
                const float r0 = ok ? bf_lo(o.x) : 0.f, r1 = ok ? bf_hi(o.x) : 0.f, r2 = ok ? bf_lo(o.y) : 0.f, r3 = ok ? bf_hi(o.y) : 0.f;
                s1[0] += r0; s1[1] += r1; s1[2] += r2; s1[3] += r3;
                s2[0] = fmaf(r0, r0, s2[0]); s2[1] = fmaf(r1, r1, s2[1]); s2[2] = fmaf(r2, r2, s2[2]); s2[3] = fmaf(r3, r3, s2[3]);
            }
        };
        auto step = [&](int r, auto phc, auto slc) {
            constexpr int SL = decltype(slc)::value;
            static_assert(PF == 5, "vmcnt below = 5 PF - 3");
            asm volatile("s_waitcnt vmcnt(22)\n\ts_barrier" ::: "memory");
            dma(r - 1 + PF, (SL + PF) % NBUF);
            compute(r, phc, slc);
        };
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int r = 0; r < PF; ++r) {
            dma(r, r);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{0u, 0u}, orsrc, OOB + 8 * (r * 2 + i), 0, 0);
            }
        }
        for (int r = 1; r <= rlast; r += 6) {
            step(r, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
            if (r + 1 > rlast) break;
            step(r + 1, std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{});
            if (r + 2 > rlast) break;
            step(r + 2, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
            if (r + 3 > rlast) break;
            step(r + 3, std::integral_constant<int, 1>{}, std::integral_constant<int, 3>{});
            if (r + 4 > rlast) break;
            step(r + 4, std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{});
            if (r + 5 > rlast) break;
            step(r + 5, std::integral_constant<int, 0>{}, std::integral_constant<int, 5>{});
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
        if (tid < 16) {
            float u = 0.f, v = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { u += red[(w * 16 + tid) * 2]; v += red[(w * 16 + tid) * 2 + 1]; }
            a.stats[((size_t)blockIdx.x * g.Cout + nt0 * 16 + tid) * 2 + 0] = u;
            a.stats[((size_t)blockIdx.x * g.Cout + nt0 * 16 + tid) * 2 + 1] = v;
        }
    }
}

int launch_conv_z32(const MfmaConvArgs& a0, hipStream_t s) {
    const ConvGeom& g = a0.g;
    if (sliding_window_off() || g.Cin != 32 || g.ks != 3 || g.stride != 1 || g.Wo < 12 || g.Do < 8 || g.D != g.Do || g.H != g.Ho || g.W != g.Wo) return 0;
    for (int k = 0; k < 2; ++k)
        if (a0.out[k] && (size_t)a0.oD * a0.oH * a0.oW * a0.outC[k] * 2 >= ((size_t)1 << 31)) return 0;
    if (a0.nout > 1 && a0.outC[0] % 16) return 0;
    for (int k = 0; k < a0.nsrc; ++k) if (a0.src[k].C % 8) return 0;
    ZWork zw;
    zw.cols_x = (g.Wo + 15) / 16; zw.cols_y = (g.Ho + 7) / 8;
    const int cols = zw.cols_x * zw.cols_y, gy = g.Cout / 16;
    int want = 512 / gy;
    if (want < 1) want = 1;
    int nseg = (want + cols - 1) / cols;
    if (nseg < 1) nseg = 1;
    int zlen = (g.Do + nseg - 1) / nseg;
    if (zlen < 4) zlen = 4;
    nseg = (g.Do + zlen - 1) / zlen;
    zw.nseg = nseg; zw.zlen = zlen;
    const int items = cols * nseg;
    const int gx = items < want ? items : want;
    constexpr int lds = 6 * 10 * 18 * 64;
    static std::atomic<uint64_t> attr_done{0};
    set_max_lds_once(attr_done, (const void*)k_mfma_conv_z32, lds);
    k_mfma_conv_z32<<<dim3((unsigned)gx, (unsigned)gy), 256, lds, s>>>(a0, zw);
    return gx;
}

bool conv_z16_applies(const MfmaConvArgs& a0) {
    const ConvGeom& g = a0.g;
    if (sliding_window_off() || g.Cin != 16 || g.ks != 3 || g.stride != 1 || g.Wo < 12 || g.Do < 8 || g.D != g.Do || g.H != g.Ho || g.W != g.Wo) return false;
    if (g.Cout != 16 && g.Cout != 32) return false;
    for (int k = 0; k < 2; ++k)   // outputs are addressed with 31-bit byte offsets through a buffer descriptor
        if (a0.out[k] && (size_t)a0.oD * a0.oH * a0.oW * a0.outC[k] * 2 >= ((size_t)1 << 31)) return false;
    if (a0.nout > 1 && a0.outC[0] % 16) return false;
    return true;
}
int launch_conv_z16(const MfmaConvArgs& a0, hipStream_t s) {
    const ConvGeom& g = a0.g;
    if (!conv_z16_applies(a0)) return 0;
    ZWork zw;
    zw.cols_x = (g.Wo + 15) / 16; zw.cols_y = (g.Ho + 7) / 8;
    const int cols = zw.cols_x * zw.cols_y;
    const int gy = g.Cout / 16;                        // one row tile per block (two row tiles: 256 VGPRs and spills)
    const int want = 512 / gy;                         // two blocks per CU in total
    int nseg = (want + cols - 1) / cols;
    if (nseg < 1) nseg = 1;
    int zlen = (g.Do + nseg - 1) / nseg;
    if (zlen < 4) zlen = 4;
    nseg = (g.Do + zlen - 1) / zlen;
    zw.nseg = nseg; zw.zlen = zlen;
    const int items = cols * nseg;
    const int gx = items < want ? items : want;
    constexpr int lds = 6 * 10 * 18 * 32;
    if (a0.bn_partial) k_mfma_conv_z16<1, true><<<dim3((unsigned)gx, (unsigned)gy), 256, lds + 6 * 8 * 16 * 32, s>>>(a0, zw);
    else k_mfma_conv_z16<1, false><<<dim3((unsigned)gx, (unsigned)gy), 256, lds, s>>>(a0, zw);
    return gx;
}

}  // namespace unet
