// MFMA implicit-GEMM conv family for gfx950 (bf16 storage, fp32 accumulate).
//
// One kernel template covers every dense contraction of the path except wgrad:
//     D[row][voxel] += W[row][k] * X[k][voxel],   k = (tap, channel),   v_mfma_f32_16x16x32_bf16
//   kind                         S  KD PAD SC   rows            taps   reads        reference
//   conv 3x3x3 s1 forward        1  3  1   -    Cout            27     input        unet.cpp:59-72
//   conv 3x3x3 s1 dgrad          1  3  1   -    Cin             27     dL/dy        (flipped, transposed filter)
//   conv 3x3x3 s2 forward        2  3  1   -    Cout            27     input
//   conv 3x3x3 s2 dgrad          1  2  0   yes  8 parities*Cin  8      dL/dy        (zero-filled parity filter)
//   conv_trans 2x2x2 s2 forward  1  1  0   yes  8 taps*Cout     1      input        unet.cpp:46-57
//   conv_trans 2x2x2 s2 dgrad    2  2  0   -    Cin             8      dL/dy
//   * A operand  = filter tile, pre-packed in fragment order (one 16-B load per lane, L2-resident).
//   * B operand  = input patches, ds_read_b128 from an LDS halo tile [HZ][HY][HX][CK channels] staged once per
//                  CK-channel chunk and re-used by all taps.
//   * sources are PLAIN bf16 tensors: a tensor with a recorded norm + activation is read through its activated copy (the
//     engine writes it once after the statistics; transforming while staging cost +85 % of the kernel).  Zero padding pads
//     the activated tensor; a channel concat {skip, x} (unet.cpp:181) is just a second source pointer.
//   * epilogue: + bias, round to bf16, 8-B stores (4 consecutive channels per lane); optional {sum, sum of squares} per
//     channel for the following norm, one partial row per persistent block; optional accumulate / two destinations (dgrad
//     of a concat); SC = depth-to-space scatter: row block `tap` goes to output voxel 2*v + tap.
//   Kernels in this file: k_mfma_conv_p (persistent halo-tile form, every kind), k_mfma_conv_z (sliding window along z for the
//   single-chunk 32-channel stride-1 layers), k_mfma_conv_small (16^3 and smaller volumes), k_conv_first_mfma (Cin = 1),
//   k_mfma_pack / k_mfma_pack_batched (fp32 torch layout -> bf16 fragments).
#include <type_traits>
#include <cstdlib>
#include "mfma_util.h"

namespace unet {

// ---- filter packing: fp32 torch layout -> bf16 fragments [chunk][kstep][row tile][lane][8] ----
enum PackMode {
    PK_CONV_FWD = 0,    // rows o = cout, k-channel i = cin, T taps:           w[(o*A + i)*T + t]              A = Cin
    PK_CONV_DGRAD = 1,  // rows o = cin,  i = cout, 27 taps flipped:           w[(i*A + o)*27 + 26 - t]        A = Cin
    PK_CONVT_DGRAD = 2, // rows o = cin,  i = cout, 8 taps:                    w[(o*B + i)*8 + t]              B = Cout
    PK_CONVT_FWD = 3,   // rows o = t*B + co, i = cin, 1 tap:                  w[(i*B + co)*8 + t]             B = Cout
    PK_CONV_S2_DGRAD = 4 // rows o = p*A + ci (p = output parity), i = cout, 8 taps k in {0,1}^3 over dy[m+k]:
                        //   per dim  p=0: k=0 -> filter tap 1 ;  p=1: k=0 -> tap 2, k=1 -> tap 0 ; else zero     A = Cin
};
// One pack unit = (channel chunk q, row tile nt): the 16 x CK x T filter values are gathered into LDS in the order that is
// contiguous in the SOURCE (the element-per-thread version read 4 B at a 108-B stride: 0.15 ms per step for 60 MB), then the
// KSTEPS fragments [lane][8] of the unit are written with 16-B stores.
constexpr int PACK_LDS_ELEMS = 16 * (32 * 27 + 1);   // bf16: the values are rounded on the way in (27 KB per block: 5 blocks per CU instead of 2)
template <int CK, int T, int MODE>   // compile-time divisors and mode: with run-time values the index arithmetic set the kernel's time
__device__ __forceinline__ void pack_unit_t(const float* __restrict__ w, __bf16* __restrict__ out, int unit, int Co, int A, int B, __bf16* lds) {
    constexpr int KSTEPS = CK == 32 ? T : (T + 1) / 2;
    const int NTT = Co / 16;
    const int q = unit / NTT, nt = unit % NTT, o0 = nt * 16;
    constexpr int RS = CK * T + 1;                   // LDS row stride (floats): [row][chan][tap]
    constexpr int n = 16 * CK * T, U = 8;            // U loads in flight per thread
    for (int e0 = threadIdx.x; e0 < n; e0 += 256 * U) {
        float v[U];
        int dst[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * 256;
            int r = 0, c = 0, t = 0;
            int64_t src = -1;
            if (e < n) {
                if constexpr (MODE == PK_CONV_FWD || MODE == PK_CONVT_DGRAD) {   // source contiguous over (chan, tap) of a row
                    const int Ax = MODE == PK_CONV_FWD ? A : B;
                    r = e / (CK * T); const int rem = e % (CK * T); c = rem / T; t = rem % T;
                    src = ((int64_t)(o0 + r) * Ax + q * CK) * T + rem;
                } else if constexpr (MODE == PK_CONV_DGRAD) {                    // source contiguous over (row, flipped tap) of a k-channel
                    c = e / (16 * 27); const int rem = e % (16 * 27); r = rem / 27; t = 26 - rem % 27;
                    src = ((int64_t)(q * CK + c) * A + o0) * 27 + rem;
                } else if constexpr (MODE == PK_CONVT_FWD) {                     // rows o = t8*B + co, one tap
                    const int t8 = o0 / B, co0 = o0 % B;
                    c = e / 16; r = e % 16; t = 0;
                    src = ((int64_t)(q * CK + c) * B + co0 + r) * 8 + t8;
                } else {                                                         // PK_CONV_S2_DGRAD: rows o = p*A + ci, taps k in {0,1}^3
                    const int pp = o0 / A, ci0 = o0 % A;
                    c = e / (16 * 8); const int rem = e % (16 * 8); r = rem / 8; t = rem % 8;
                    const int pk[3] = {(pp >> 2) & 1, (pp >> 1) & 1, pp & 1}, kk[3] = {(t >> 2) & 1, (t >> 1) & 1, t & 1};
                    int ft[3];
                    bool ok = true;
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        if (pk[d] == 0) { ok = ok && kk[d] == 0; ft[d] = 1; }
                        else ft[d] = kk[d] == 0 ? 2 : 0;
                    }
                    if (ok) src = ((int64_t)(q * CK + c) * A + ci0 + r) * 27 + (ft[0] * 9 + ft[1] * 3 + ft[2]);
                }
            }
            dst[u] = e < n ? r * RS + c * T + t : -1;
            v[u] = src >= 0 ? w[src] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (dst[u] >= 0) lds[dst[u]] = (__bf16)v[u];
    }
    __syncthreads();
    for (int f = threadIdx.x; f < KSTEPS * 64; f += 256) {
        const int ks = f >> 6, lane = f & 63, row = lane & 15;
        int tap, c0;
        if (CK == 32) { tap = ks; c0 = 8 * (lane >> 4); }
        else { tap = 2 * ks + (lane >> 5); c0 = 8 * ((lane >> 4) & 1); }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = tap < T ? lds[row * RS + (c0 + e) * T + tap] : (__bf16)0.f;
        *(bf16x8*)(out + ((((int64_t)q * KSTEPS + ks) * NTT + nt) * 64 + lane) * 8) = o;
    }
}
__device__ __forceinline__ void pack_unit(const float* __restrict__ w, __bf16* __restrict__ out, int unit, int Ci, int Co, int CK, int T,
                                          int mode, int A, int B, __bf16* lds) {
    switch (mode) {
        case PK_CONV_FWD:
            if (CK == 32) pack_unit_t<32, 27, PK_CONV_FWD>(w, out, unit, Co, A, B, lds); else pack_unit_t<16, 27, PK_CONV_FWD>(w, out, unit, Co, A, B, lds);
            break;
        case PK_CONV_DGRAD:
            if (CK == 32) pack_unit_t<32, 27, PK_CONV_DGRAD>(w, out, unit, Co, A, B, lds); else pack_unit_t<16, 27, PK_CONV_DGRAD>(w, out, unit, Co, A, B, lds);
            break;
        case PK_CONVT_DGRAD:
            if (CK == 32) pack_unit_t<32, 8, PK_CONVT_DGRAD>(w, out, unit, Co, A, B, lds); else pack_unit_t<16, 8, PK_CONVT_DGRAD>(w, out, unit, Co, A, B, lds);
            break;
        case PK_CONVT_FWD: pack_unit_t<32, 1, PK_CONVT_FWD>(w, out, unit, Co, A, B, lds); break;
        default: pack_unit_t<32, 8, PK_CONV_S2_DGRAD>(w, out, unit, Co, A, B, lds); break;
    }
}
__global__ void __launch_bounds__(256) k_mfma_pack(const float* __restrict__ w, __bf16* __restrict__ out, int Ci, int Co, int CK, int T,
                                                   int mode, int A, int B) {
    __shared__ __bf16 lds[PACK_LDS_ELEMS];
    pack_unit(w, out, blockIdx.x, Ci, Co, CK, T, mode, A, B, lds);
}
// every filter pack of a plan in ONE launch: block -> (job, unit) by binary search over the jobs' first block index.  The launch serves
// the pack units [blk_base, blk_base + nunits) of the job table; with fewer blocks than units a block walks units blockIdx.x,
// blockIdx.x + gridDim.x, ... (the deep levels' packs run on the side stream beside the forward's first convs and are not needed
// before the caller's stream reaches those levels: a bounded grid leaves the CUs' LDS and wave slots to the convs).
__global__ void __launch_bounds__(256) k_mfma_pack_batched(const float* __restrict__ params_base, char* __restrict__ ws,
                                                           const PackJob* __restrict__ jobs, int njobs, int64_t blk_base, int64_t nunits,
                                                           int* __restrict__ zero, int nzero) {
    __shared__ __bf16 lds[PACK_LDS_ELEMS];
    // the first pack launch of a forward also clears the deep levels' arrival counters (kernels_mfma_deep.hip): every consumer of a
    // counter waits for this launch anyway (it reads a filter pack), so the workspace needs no separate initialisation
    if (zero && blockIdx.x == 0)
        for (int i = threadIdx.x; i < nzero; i += 256) zero[i] = 0;
    for (int64_t u = blockIdx.x; u < nunits; u += gridDim.x) {
        const int64_t blk = blk_base + u;
        int lo = 0, hi = njobs - 1;
        while (lo < hi) {
            int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].blk0 <= blk) lo = mid; else hi = mid - 1;
        }
        const PackJob jb = jobs[lo];
        pack_unit(params_base + jb.src_off, (__bf16*)(ws + jb.dst_off), (int)(blk - jb.blk0), jb.Ci, jb.Co, jb.CK, jb.T,
                  jb.mode, jb.A, jb.B, lds);
        __syncthreads();                              // the unit's fragments are read from LDS before the next unit overwrites it
    }
}
void launch_mfma_pack_batched(const float* params_base, void* ws, const PackJob* jobs_dev, int njobs, int64_t nblocks, hipStream_t s,
                              int64_t blk_base, int max_grid, int* zero, int nzero) {
    if (njobs <= 0 || nblocks <= 0) return;
    const int64_t grid = (max_grid > 0 && nblocks > max_grid) ? max_grid : nblocks;
    k_mfma_pack_batched<<<(unsigned)grid, 256, 0, s>>>(params_base, (char*)ws, jobs_dev, njobs, blk_base, nblocks, zero, nzero);
}

static inline int pick_ck(int Ci, bool allow32) { return (allow32 && Ci % 32 == 0) ? 32 : 16; }
static size_t pack_bytes(int Ci, int Co, int CK, int T) {
    int KSTEPS = CK == 32 ? T : (T + 1) / 2;
    return (size_t)(Ci / CK) * KSTEPS * (Co / 16) * 64 * 16;
}
static PackJob make_job(int Ci, int Co, int CK, int T, int mode, int A, int B) {
    PackJob j;
    j.src_off = 0; j.dst_off = 0; j.blk0 = 0;
    j.Ci = Ci; j.Co = Co; j.CK = CK; j.T = T; j.mode = mode; j.A = A; j.B = B; j.pad = 0;
    j.total = (int64_t)(Ci / CK) * (Co / 16);      // pack units = blocks of the batched launch
    return j;
}
static void run_pack(const float* w, void* out, int Ci, int Co, int CK, int T, int mode, int A, int B, hipStream_t s) {
    k_mfma_pack<<<(unsigned)((Ci / CK) * (Co / 16)), 256, 0, s>>>(w, (__bf16*)out, Ci, Co, CK, T, mode, A, B);
}

// ------------------------------------------------------------------------------------------------
// Persistent, software-pipelined kernel.  A block walks
// tiles bid, bid + gridDim.x, ...; one pipeline stage = one (tile, channel chunk).  The global loads of
// stage s+1 are issued into registers BEFORE the MFMAs of stage s and written to LDS after them, so HBM/L2
// latency hides under the matrix work even at 2 blocks per CU (measured: the first, one-tile-per-block version spent ~6 % of
// a block's lifetime in MFMAs, the rest waiting for the halo tile).
// ------------------------------------------------------------------------------------------------
// ONE: Cin == CK is known at compile time (the two heaviest layers, 32->16 and 16->16 at 128^3): the filter fragments are
// staged once before the loop instead of riding in prefetch registers through every stage (those 16 VGPRs made the
// 128-VGPR variants spill their prefetched halo data), and the stage -> (tile, chunk) divisions disappear.
template <int S, int KD, int PAD, int BZ, int BY, int BX, int CK, int NT, bool SC, int NW, bool ONE>
// launch bounds: 2 resident blocks per CU (4 waves/SIMD, <= 128 VGPRs) for the 8-wave light configurations, else 2 waves/SIMD (<= 256)
__global__ void __launch_bounds__(NW * 64, (NW == 8 && NT == 1) ? 4 : 2) k_mfma_conv_p(MfmaConvArgs a) {
    constexpr int NTHR = NW * 64;   // 8 waves on the larger tiles: half the staging registers per thread, 4 waves per SIMD at 2 blocks/CU
    constexpr int HZ = (BZ - 1) * S + KD, HY = (BY - 1) * S + KD, HX = (BX - 1) * S + KD, NVOX = HZ * HY * HX;
    constexpr int G = CK / 8;
    // 3x3x3 stride-1 tiles with 64-B voxels: unpadded 64-B voxels, rows padded to a multiple of 4 voxels (row starts on a
    // 256-B bank row) and the 16-B group index XOR-ed with bit 2 of the voxel's x (swz): conflict-free ds_read_b128 for
    // 16 consecutive x at any tap shift (checked exhaustively offline), 46 KB instead of 62 KB per tile -> 3 blocks/CU.
    constexpr bool SWZ = CK == 32 && S == 1 && KD == 3;
    constexpr int VS = SWZ ? 64 : (CK == 32 ? 96 : 32);
    constexpr int HXP = SWZ ? (HX + 3) / 4 * 4 : HX;   // row pitch in voxels
    constexpr int TXM = BX < 16 ? BX : 16;
    constexpr int TYM = 16 / TXM;
    constexpr int MT = BZ * BY * BX / 16, MTW = MT / NW;
    constexpr int T = KD * KD * KD;
    constexpr int KSTEPS = CK == 32 ? T : (T + 1) / 2;
    constexpr int UNITS = NVOX * G, ITERS = (UNITS + NTHR - 1) / NTHR;
    constexpr int NKX = SWZ ? 3 : 1;
    // The chunk's filter fragments live in LDS behind the tile when they fit (<= 32 KB): in the ISA of the first version
    // every tap waited on a global (L2) load issued one tap earlier -- ~27 exposed L2 latencies per tile, the whole
    // difference between ~6 % and the MFMA-issue-limited rate.  With Cin <= CK they are loaded once per block.
    constexpr bool WLDS = KSTEPS * NT <= 32;
    constexpr int TILE_B = HZ * HY * HXP * VS;
    constexpr int WPIECES = KSTEPS * NT * 64, WITERS = WLDS ? (WPIECES + NTHR - 1) / NTHR : 1;
    static_assert(MT % NW == 0 && MTW >= 1, "tile must give every wave at least one m-tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, gq = lane >> 4;
    const int nblk = a.tiles_x * a.tiles_y * a.tiles_z;
    const int nt0 = blockIdx.y * NT, NTT = g.Cout / 16;
    const int C0 = a.src[0].C;
    const int lg = tid % G;
    const int nchunk = ONE ? 1 : g.Cin / CK;
    const bf16x8* wp = (const bf16x8*)a.w;

    int mbase[MTW][NKX];
    unsigned mvox[MTW];                 // this lane's voxel of m-tile i: offset from the tile origin in the destination volume (non-scatter kinds)
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        int mt = wave * MTW + i;
        constexpr int RG = BY / TYM;
        const int mz = mt / RG, my = (mt % RG) * TYM + (j / TXM), mx = j % TXM;
        mvox[i] = (unsigned)((mz * a.oH + my) * a.oW + mx);
        if (SWZ) {
#pragma unroll
            for (int kx = 0; kx < NKX; ++kx) {
                const int hx = mx + kx;
                mbase[i][kx] = ((mz * HY + my) * HXP + hx) * 64 + ((gq ^ (((hx >> 2) & 1) << 1)) << 4);
            }
        } else {
            mbase[i][0] = ((mz * S * HY + my * S) * HX + mx * S) * VS + (CK == 32 ? gq : (gq & 1)) * 16;
        }
    }
    f32x4 acc[MTW][NT];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // my tiles: k-th tile of this block is xcd_remap(blockIdx.x + k * gridDim.x)
    const int my_tiles = nblk > (int)blockIdx.x ? (nblk - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int nst = my_tiles * nchunk;

    // Per-thread staging units, decoded ONCE: halo coordinates (packed 10 bits each, -1 = no unit) and LDS byte offset.
    // (PMC on the first version: 39 % of wave cycles issuing instructions, almost all of it this index arithmetic
    // repeated per unit per tile: constant divisions, 64-bit address math, swizzle.)
    // uvox = the unit's voxel index relative to the halo origin in the SOURCE volume: a load address is
    // base(tile, chunk) + uvox * voxel stride, one 24-bit multiply-add per unit (the 64-bit (z*H+y)*W+x chains of the
    // previous version were ~100 quarter-rate VALU instructions per tile and wave -- 6 VALU per MFMA, PMC SQ_INSTS_VALU).
    // upk = halo coordinates (z 4 bits | y 5 bits << 4 | x 6 bits << 9) | LDS offset / 16 << 15, or -1 for "no unit": one register
    static_assert(HZ <= 16 && HY <= 32 && HX <= 64 && TILE_B / 16 < 65536, "staging unit packing");
    int upk[ITERS];
    constexpr bool TIGHT = NW == 8 && NT == 1;     // the 128-VGPR variants recompute uvox from upk (5 VALU) instead of holding it
    unsigned uvox[TIGHT ? 1 : ITERS];
    const int HW = g.H * g.W;
    auto uvox_of = [&](int it) -> unsigned {
        if constexpr (!TIGHT) return uvox[it];
        else { const int uc = upk[it]; return (unsigned)(__umul24(uc & 15, HW) + __umul24((uc >> 4) & 31, g.W) + ((uc >> 9) & 63)); }
    };
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int u = tid + it * NTHR;
        const int hv = u / G, hz = hv / (HY * HX), hr = hv % (HY * HX), hy = hr / HX, hx = hr % HX;
        const int lds_off = SWZ ? ((hz * HY + hy) * HXP + hx) * 64 + ((lg ^ (((hx >> 2) & 1) << 1)) << 4) : hv * VS + lg * 16;
        upk[it] = u < UNITS ? (hz | (hy << 4) | (hx << 9) | ((lds_off >> 4) << 15)) : -1;
        if constexpr (!TIGHT) uvox[it] = (unsigned)((hz * g.H + hy) * g.W + hx);
    }
    // RW is a native vector type: with the HIP uint4 struct the copies became llvm.memcpy global->private->LDS that SROA left in
    // scratch memory (global_load, s_waitcnt, scratch_store per piece in every stage's prefetch)
    uint4 R[ITERS];
    bf16x8 RW[WITERS];
    // issue the global loads of stage st into R (no waiting)
    auto prefetch = [&](int st) {
        const int k = st / nchunk, q = st - k * nchunk;
        if (!ONE && WLDS && (nchunk > 1 || st == 0)) {
#pragma unroll
            for (int it = 0; it < WITERS; ++it) {
                const int pc = tid + it * NTHR;         // piece = (ks, n, lane)
                if (pc < WPIECES) {
                    const int l = pc & 63, kn = pc >> 6, n = kn % NT, ks = kn / NT;
                    RW[it] = wp[(((size_t)q * KSTEPS + ks) * NTT + nt0 + n) * 64 + l];
                }
            }
        }
        const int bid = xcd_remap((int)blockIdx.x + k * (int)gridDim.x, nblk);
        const int x0 = (bid % a.tiles_x) * BX, y0 = ((bid / a.tiles_x) % a.tiles_y) * BY, z0 = (bid / (a.tiles_x * a.tiles_y)) * BZ;
        const int iz0 = z0 * S - PAD, iy0 = y0 * S - PAD, ix0 = x0 * S - PAD;
        const int c = q * CK + lg * 8;
        const int s = (a.nsrc > 1 && c >= C0) ? 1 : 0;
        // (selects, not a.src[s]: indexing a kernel argument dynamically makes the compiler copy it to scratch memory)
        const char* sptr = (const char*)(s ? a.src[1].ptr : a.src[0].ptr);
        const int sC = s ? a.src[1].C : C0;
        const long long org = ((long long)iz0 * g.H + iy0) * g.W + ix0;          // halo origin (may lie outside the volume)
        const char* base = sptr + (org * sC + (c - (s ? C0 : 0))) * 2;
        const unsigned vstride = (unsigned)sC * 2;
        const bool interior = iz0 >= 0 && iy0 >= 0 && ix0 >= 0 && iz0 + HZ <= g.D && iy0 + HY <= g.H && ix0 + HX <= g.W;
        if (interior) {
#pragma unroll
            for (int it = 0; it < ITERS; ++it)
                if ((it + 1) * NTHR <= UNITS || upk[it] >= 0) R[it] = *(const uint4*)(base + __umul24(uvox_of(it), vstride));
        } else {
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int uc = upk[it];
                const int gz = iz0 + (uc & 15), gy = iy0 + ((uc >> 4) & 31), gx = ix0 + ((uc >> 9) & 63);
                R[it] = make_uint4(0u, 0u, 0u, 0u);
                if (uc >= 0 && (unsigned)gz < (unsigned)g.D && (unsigned)gy < (unsigned)g.H && (unsigned)gx < (unsigned)g.W)
                    R[it] = *(const uint4*)(base + __umul24(uvox_of(it), vstride));
            }
        }
    };
    // write R to the LDS tile (sources are plain: a tensor with a pending norm/activation is read through its activated copy)
    auto commit = [&](int st) {
        const int k = st / nchunk, q = st - k * nchunk;
        if (!ONE && WLDS && (nchunk > 1 || st == 0)) {
#pragma unroll
            for (int it = 0; it < WITERS; ++it) {
                const int pc = tid + it * NTHR;
                if (pc < WPIECES) *(bf16x8*)(smem + TILE_B + pc * 16) = RW[it];
            }
        }
#pragma unroll
        for (int it = 0; it < ITERS; ++it)
            if ((it + 1) * NTHR <= UNITS || upk[it] >= 0) *(uint4*)(smem + ((upk[it] >> 15) << 4)) = R[it];
    };

    float s1[NT][4], s2[NT][4];      // per-thread norm statistics of the stored outputs (channels nt0*16 + n*16 + gq*4 + r)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[n][r] = 0.f; s2[n][r] = 0.f; }

    if (ONE && WLDS) {   // the only chunk's filter fragments: global -> LDS once (visible after the first stage's barrier)
#pragma unroll
        for (int it = 0; it < WITERS; ++it) {
            const int pc = tid + it * NTHR;
            if (pc < WPIECES) {
                const int l = pc & 63, kn = pc >> 6, n = kn % NT, ks = kn / NT;
                *(bf16x8*)(smem + TILE_B + pc * 16) = wp[((size_t)ks * NTT + nt0 + n) * 64 + l];
            }
        }
    }
    if (nst > 0) prefetch(0);
    for (int st = 0; st < nst; ++st) {
        const int k = st / nchunk, q = st - k * nchunk;
        const bf16x8* wq = wp + ((size_t)q * KSTEPS * NTT + nt0) * 64 + lane;
        __syncthreads();                 // every wave is done reading the previous stage's tile
        commit(st);
        __syncthreads();
        if (st + 1 < nst) prefetch(st + 1);   // in flight during the MFMAs below

        // Tap loop, software-pipelined by hand: the LDS reads (patches + filter fragments) of tap ks+1 are issued before
        // the MFMAs of tap ks and the order is pinned with sched_barrier -- left to itself hipcc placed every ds_read one
        // instruction ahead of its MFMA (s_waitcnt lgkmcnt(1) per MFMA: one LDS latency per 16-cycle MFMA).
        auto tap_off = [&](int ks, int& toff, int& kxs) {
            kxs = 0;
            if (SWZ) { toff = ((ks / 9) * HY + (ks / 3) % 3) * HXP * 64; kxs = ks % 3; }
            else if (CK == 32) toff = (((ks / (KD * KD)) * HY + (ks / KD) % KD) * HX + ks % KD) * VS;
            else {
                const int t0 = 2 * ks, t1 = 2 * ks + 1 < T ? 2 * ks + 1 : 2 * ks;
                const int o0 = (((t0 / (KD * KD)) * HY + (t0 / KD) % KD) * HX + t0 % KD) * VS;
                const int o1 = (((t1 / (KD * KD)) * HY + (t1 / KD) % KD) * HX + t1 % KD) * VS;
                toff = (lane & 32) ? o1 : o0;
            }
        };
        // PD taps in flight ahead of the MFMAs: 2 where the register budget allows (4-wave single-chunk blocks: 2 waves per SIMD
        // leave an LDS latency exposed with 1)
        constexpr int PD = 1;   // measured: PD 2 on the 4-wave single-chunk blocks left the 32->16 layer at 0.082 ms and spilled the 16->16 one
        bf16x8 xbuf[PD + 1][MTW], wbuf[PD + 1][NT];
        auto load_tap = [&](int ks, int slot) {
            int toff, kxs;
            tap_off(ks, toff, kxs);
#pragma unroll
            for (int n = 0; n < NT; ++n)
                wbuf[slot][n] = WLDS ? *(const bf16x8*)(smem + TILE_B + ((ks * NT + n) * 64 + lane) * 16) : wq[((size_t)ks * NTT + n) * 64];
#pragma unroll
            for (int i = 0; i < MTW; ++i) xbuf[slot][i] = *(const bf16x8*)(smem + mbase[i][SWZ ? kxs : 0] + toff);
        };
#pragma unroll
        for (int k0 = 0; k0 < PD; ++k0)
            if (k0 < KSTEPS) load_tap(k0, k0);
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            if (ks + PD < KSTEPS) load_tap(ks + PD, (ks + PD) % (PD + 1));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MTW; ++i)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wbuf[ks % (PD + 1)][n], xbuf[ks % (PD + 1)][i], acc[i][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (q != nchunk - 1) continue;

        // ---- tile epilogue ----
        const int bid = xcd_remap((int)blockIdx.x + k * (int)gridDim.x, nblk);
        const int x0 = (bid % a.tiles_x) * BX, y0 = ((bid / a.tiles_x) % a.tiles_y) * BY, z0 = (bid / (a.tiles_x * a.tiles_y)) * BZ;
        if constexpr (SC && KD == 2) {      // (the stride-2 dgrad; conv_trans forward, KD = 1, never accumulates: it keeps the plain loop and 46 VGPRs)
            // scatter kinds that ACCUMULATE (the stride-2 dgrad adds to the skip tensor's gradient): all old values are requested
            // before the first one is used.  In the generic loop below every (m-tile, row tile) is load -> wait -> add -> store, and the
            // compiler cannot move a load above the previous store: eight exposed latencies per tile (77 us against 41 us write-only).
            bool any_acc = false;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int c = (nt0 + n) * 16 + gq * 4 - ((nt0 + n) * 16 + gq * 4) / a.sc_C * a.sc_C;
                any_acc = any_acc || ((a.nout > 1 && c >= a.outC[0]) ? a.out_acc[1] : a.out_acc[0]);
            }
            if (any_acc) {
                uint2* pp[MTW][NT];
                uint2 old[MTW][NT];
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    int c = (nt0 + n) * 16 + gq * 4;
                    const int tap = c / a.sc_C;
                    c -= tap * a.sc_C;
                    const int tz = tap >> 2, ty = (tap >> 1) & 1, tx = tap & 1;
                    const int d = (a.nout > 1 && c >= a.outC[0]) ? 1 : 0, cd = c - (d ? a.outC[0] : 0);
                    char* obase = (char*)(d ? a.out[1] : a.out[0]);
                    const int oC = d ? a.outC[1] : a.outC[0], oacc = d ? a.out_acc[1] : a.out_acc[0];
#pragma unroll
                    for (int i = 0; i < MTW; ++i) {
                        const int mt = wave * MTW + i;
                        const int gz = 2 * (z0 + mt / (BY / TYM)) + tz, gy = 2 * (y0 + (mt % (BY / TYM)) * TYM + (j / TXM)) + ty, gx = 2 * (x0 + j % TXM) + tx;
                        pp[i][n] = (gz < a.oD && gy < a.oH && gx < a.oW && obase) ? (uint2*)(obase + ((((size_t)gz * a.oH + gy) * a.oW + gx) * oC + cd) * 2) : nullptr;
                        old[i][n] = (pp[i][n] && oacc) ? *pp[i][n] : make_uint2(0u, 0u);
                    }
                }
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    int c = (nt0 + n) * 16 + gq * 4;
                    c -= c / a.sc_C * a.sc_C;
                    float b4[4] = {0.f, 0.f, 0.f, 0.f};
                    if (a.bias) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) b4[r] = a.bias[c + r];
                    }
#pragma unroll
                    for (int i = 0; i < MTW; ++i) {
                        if (pp[i][n]) {
                            uint2 o;
                            o.x = pack_bf16x2(acc[i][n][0] + b4[0] + bf_lo(old[i][n].x), acc[i][n][1] + b4[1] + bf_hi(old[i][n].x));
                            o.y = pack_bf16x2(acc[i][n][2] + b4[2] + bf_lo(old[i][n].y), acc[i][n][3] + b4[3] + bf_hi(old[i][n].y));
                            *pp[i][n] = o;
                        }
                        acc[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
                continue;
            }
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            int c = (nt0 + n) * 16 + gq * 4;
            int tz = 0, ty = 0, tx = 0;
            if (SC) { int tap = c / a.sc_C; c -= tap * a.sc_C; tz = tap >> 2; ty = (tap >> 1) & 1; tx = tap & 1; }
            float b4[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.bias) {
#pragma unroll
                for (int r = 0; r < 4; ++r) b4[r] = a.bias[c + r];
            }
            int d = (a.nout > 1 && c >= a.outC[0]) ? 1 : 0;
            int cd = c - (d ? a.outC[0] : 0);
            char* obase = (char*)(d ? a.out[1] : a.out[0]);
            const int oC = d ? a.outC[1] : a.outC[0], oacc = d ? a.out_acc[1] : a.out_acc[0];
            // non-scatter kinds: address = tile origin (uniform) + this lane's precomputed voxel offset
            const bool edge = z0 + BZ > a.oD || y0 + BY > a.oH || x0 + BX > a.oW;
            char* tbase = obase + ((((long long)z0 * a.oH + y0) * a.oW + x0) * oC + cd) * 2;
#pragma unroll
            for (int i = 0; i < MTW; ++i) {
                const int mt = wave * MTW + i;    // recomputed (cheap, constant divisors) rather than held in registers
                int gz = z0 + mt / (BY / TYM), gy = y0 + (mt % (BY / TYM)) * TYM + (j / TXM), gx = x0 + j % TXM;
                if (SC) { gz = 2 * gz + tz; gy = 2 * gy + ty; gx = 2 * gx + tx; }
                if (((!SC && !edge) || (gz < a.oD && gy < a.oH && gx < a.oW)) && obase) {
                    uint2* p;
                    if (SC) p = (uint2*)(obase + ((((size_t)gz * a.oH + gy) * a.oW + gx) * oC + cd) * 2);
                    else p = (uint2*)(tbase + __umul24(mvox[i], (unsigned)oC * 2));
                    float v0 = acc[i][n][0] + b4[0], v1 = acc[i][n][1] + b4[1], v2 = acc[i][n][2] + b4[2], v3 = acc[i][n][3] + b4[3];
                    if (oacc) {
                        uint2 old = *p;
                        v0 += bf_lo(old.x); v1 += bf_hi(old.x); v2 += bf_lo(old.y); v3 += bf_hi(old.y);
                    }
                    uint2 o;
                    o.x = pack_bf16x2(v0, v1); o.y = pack_bf16x2(v2, v3);
                    *p = o;
                    if (!SC && a.stats) {
                        float r0 = bf_lo(o.x), r1 = bf_hi(o.x), r2 = bf_lo(o.y), r3 = bf_hi(o.y);
                        s1[n][0] += r0; s1[n][1] += r1; s1[n][2] += r2; s1[n][3] += r3;
                        s2[n][0] = fmaf(r0, r0, s2[n][0]); s2[n][1] = fmaf(r1, r1, s2[n][1]);
                        s2[n][2] = fmaf(r2, r2, s2[n][2]); s2[n][3] = fmaf(r3, r3, s2[n][3]);
                    }
                }
                acc[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    }
    // norm statistics: every thread summed its own channels over all of the block's tiles; one reduction per block,
    // one partial row per persistent block (row = blockIdx.x, gridDim.x rows in all)
    if (!SC && a.stats) {
        float* red = (float*)smem;
        __syncthreads();
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float u = s1[n][r], v = s2[n][r];
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) { u += __shfl_xor(u, m); v += __shfl_xor(v, m); }
                if (j == 0) {
                    int cl = n * 16 + gq * 4 + r;
                    red[(wave * NT * 16 + cl) * 2 + 0] = u;
                    red[(wave * NT * 16 + cl) * 2 + 1] = v;
                }
            }
        __syncthreads();
        if (tid < NT * 16) {
            float u = 0.f, v = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) { u += red[(w * NT * 16 + tid) * 2]; v += red[(w * NT * 16 + tid) * 2 + 1]; }
            int c = nt0 * 16 + tid;
            a.stats[((size_t)blockIdx.x * g.Cout + c) * 2 + 0] = u;
            a.stats[((size_t)blockIdx.x * g.Cout + c) * 2 + 1] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Small volumes (the 16^3 / 8^3 / 4^3 levels of the default architecture: 3 % of the FLOPs but 15 % of the step in the
// persistent kernel).  With 64..512 blocks of tiny tiles every pipeline stage of k_mfma_conv_p is one exposed latency:
// (global load -> LDS -> barrier -> 27 LDS-latency-bound taps) x Cin/32 chunks, ~6 us per chunk at one wave per SIMD.
// Here a block (4 waves) owns a 64-voxel tile x 16 output rows and
//   1. stages the halo tile of up to SMALL_QS = 8 channel chunks (256 channels, <= 147 KB) in ONE burst of loads,
//   2. splits the chunks over its waves (wave w: chunks w, w+4): every wave runs all 4 m-tiles of the tile over its share
//      of K with the chunk's 27 filter fragments held in registers (loaded straight from L2 in lane order, refilled for
//      the wave's next chunk as the taps retire them), so the tap loop waits on LDS only,
//   3. sums the 4 partial accumulators through LDS; wave w finishes m-tile w (bias, bf16, statistics, accumulate).
// Same LDS tile layout (64-B voxels, swizzled), filter pack, arguments and epilogue semantics as k_mfma_conv_p<1,3,1,..,32,1,false>.
// ------------------------------------------------------------------------------------------------
// BNS (dgrad only, as k_mfma_conv_z16<.., true>): the destination is the view of a norm layer read by this conv alone, so its gradient is
// complete when this kernel has written it: the epilogue also reads the RAW tensor at the voxels it stores and leaves the norm backward's
// statistics {sum dv, sum dv * xhat}, dv = dL/d(view) * act'(u * scale + shift), as one row per tile in a.bn_partial -- the separate
// k_norm_bwd_stats8 launch (5-6 us of the caller's stream per layer at these levels) is not needed.
constexpr int SMALL_QS = 8;
// NW = 8 (layers of 256+ input channels): one channel chunk per wave per super-stage instead of two -- the tap loop, the longest link of
// the block's latency chain after the launch itself, is half as long; waves 4..7 only contribute their K partials.
template <int BZ, int BY, int BX, int OCC, bool BNS = false, int NW = 4>
__global__ void __launch_bounds__(NW * 64, OCC) k_mfma_conv_small(MfmaConvArgs a) {
    constexpr int NTHR = NW * 64;
    constexpr int HZ = BZ + 2, HY = BY + 2, HX = BX + 2, HXP = (HX + 3) / 4 * 4;
    constexpr int TXM = BX < 16 ? BX : 16, TYM = 16 / TXM, RG = BY / TYM;
    constexpr int TILE_B = HZ * HY * HXP * 64;
    constexpr int UNITS = HZ * HY * HX * 4, ITERS = (UNITS + NTHR - 1) / NTHR;
    constexpr int QB = OCC == 1 ? 4 : 2;   // chunks per staging burst (registers: QB * ITERS * 4)
    static_assert(BZ * BY * BX == 64, "four m-tiles per tile");
    static_assert(NW == 4 || (NW == 8 && OCC == 1), "eight waves: the one-block-per-CU form only");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, gq = lane >> 4, lg = tid & 3;
    // grid = (row tiles, voxel tiles): workgroups go to the 8 XCDs round-robin in linear order, so blocks of one XCD share their ROW TILE
    // (the 110-440 KB filter slice is fetched into one L2) and differ in the voxel tile.  With the voxel tile in x every XCD held one
    // tile and streamed the WHOLE filter: 8 x 3.5 MB per launch at 8^3 (profiles/r20_step_hbm_traffic_per_kernel.txt: 295 MB per step in
    // 11 launches against 29 MB of filters).
    const int nt0 = blockIdx.x, NTT = g.Cout / 16, nchunk = g.Cin / 32, C0 = a.src[0].C;
    const int bid = blockIdx.y;
    const int x0 = (bid % a.tiles_x) * BX, y0 = ((bid / a.tiles_x) % a.tiles_y) * BY, z0 = (bid / (a.tiles_x * a.tiles_y)) * BZ;
    const int iz0 = z0 - 1, iy0 = y0 - 1, ix0 = x0 - 1;
    const bf16x8* wp = (const bf16x8*)a.w;

    int mbase[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int mz = i / RG, my = (i % RG) * TYM + j / TXM, mx = j % TXM;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int hx = mx + kx;
            mbase[i][kx] = ((mz * HY + my) * HXP + hx) * 64 + ((gq ^ (((hx >> 2) & 1) << 1)) << 4);
        }
    }
    int ulds[ITERS];
    unsigned uvox[ITERS], uin = 0;            // uin bit it: the unit exists and lies inside the volume
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int u = tid + it * NTHR;
        const int hv = u >> 2, hz = hv / (HY * HX), hr = hv % (HY * HX), hy = hr / HX, hx = hr % HX;
        ulds[it] = u < UNITS ? ((hz * HY + hy) * HXP + hx) * 64 + ((lg ^ (((hx >> 2) & 1) << 1)) << 4) : -1;
        uvox[it] = (unsigned)((hz * g.H + hy) * g.W + hx);
        const int gz = iz0 + hz, gy = iy0 + hy, gx = ix0 + hx;
        if (u < UNITS && (unsigned)gz < (unsigned)g.D && (unsigned)gy < (unsigned)g.H && (unsigned)gx < (unsigned)g.W) uin |= 1u << it;
    }
    const long long org = ((long long)iz0 * g.H + iy0) * g.W + ix0;
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int q0 = 0; q0 < nchunk; q0 += SMALL_QS) {
        const int nq = nchunk - q0 < SMALL_QS ? nchunk - q0 : SMALL_QS;
        // this wave's first chunk of the super-stage: its 27 filter fragments, in flight during the staging below
        bf16x8 wf[27];
        {
            const int qq = wave < nq ? wave : 0;
            const bf16x8* wq = wp + ((size_t)(q0 + qq) * 27 * NTT + nt0) * 64 + lane;
#pragma unroll
            for (int ks = 0; ks < 27; ++ks) wf[ks] = wq[(size_t)ks * NTT * 64];
        }
        __syncthreads();                 // the previous super-stage's tiles are no longer read
#pragma unroll 1
        for (int qb = 0; qb < nq; qb += QB) {
            uint4 R[QB][ITERS];
#pragma unroll
            for (int qq = 0; qq < QB; ++qq) {
                if (qb + qq < nq) {
                    const int c = (q0 + qb + qq) * 32 + lg * 8;
                    const int sidx = (a.nsrc > 1 && c >= C0) ? 1 : 0;
                    const char* sptr = (const char*)(sidx ? a.src[1].ptr : a.src[0].ptr);
                    const int sC = sidx ? a.src[1].C : C0;
                    const char* base = sptr + (org * sC + (c - (sidx ? C0 : 0))) * 2;
                    const unsigned vstride = (unsigned)sC * 2;
#pragma unroll
                    for (int it = 0; it < ITERS; ++it) {
                        R[qq][it] = make_uint4(0u, 0u, 0u, 0u);
                        if ((uin >> it) & 1u) R[qq][it] = *(const uint4*)(base + __umul24(uvox[it], vstride));
                    }
                }
            }
#pragma unroll
            for (int qq = 0; qq < QB; ++qq) {
                if (qb + qq < nq) {
#pragma unroll
                    for (int it = 0; it < ITERS; ++it)
                        if (ulds[it] >= 0) *(uint4*)(smem + (qb + qq) * TILE_B + ulds[it]) = R[qq][it];
                }
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int qq = wave; qq < nq; qq += NW) {
            const char* tile = smem + qq * TILE_B;
            const bool more = OCC == 1 && qq + NW < nq;   // OCC 2 is launched with nq <= 4 only: one chunk per wave
            const bf16x8* wn = wp + ((size_t)(q0 + qq + NW) * 27 * NTT + nt0) * 64 + lane;   // the wave's next chunk (if more)
            bf16x8 xbuf[2][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) xbuf[0][i] = *(const bf16x8*)(tile + mbase[i][0]);
#pragma unroll
            for (int ks = 0; ks < 27; ++ks) {
                if (ks + 1 < 27) {
                    const int k1 = ks + 1, toff = ((k1 / 9) * HY + (k1 / 3) % 3) * HXP * 64;
#pragma unroll
                    for (int i = 0; i < 4; ++i) xbuf[k1 & 1][i] = *(const bf16x8*)(tile + mbase[i][k1 % 3] + toff);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks], xbuf[ks & 1][i], acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (more) wf[ks] = wn[(size_t)ks * NTT * 64];
            }
        }
    }

    // ---- sum the NW K-partials; wave w < 4 finishes m-tile w ----
    __syncthreads();
    float* red = (float*)smem;
#pragma unroll
    for (int i = 0; i < 4; ++i) *(f32x4*)(red + ((wave * 4 + i) * 64 + lane) * 4) = acc[i];
    __syncthreads();
    const int mt = wave & 3;
    f32x4 o4 = *(const f32x4*)(red + ((0 * 4 + mt) * 64 + lane) * 4);
#pragma unroll
    for (int v = 1; v < NW; ++v) {
        const f32x4 t = *(const f32x4*)(red + ((v * 4 + mt) * 64 + lane) * 4);
        o4[0] += t[0]; o4[1] += t[1]; o4[2] += t[2]; o4[3] += t[3];
    }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    {
        const int c = nt0 * 16 + gq * 4;
        float b4[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r) b4[r] = a.bias[c + r];
        }
        const int d = (a.nout > 1 && c >= a.outC[0]) ? 1 : 0;
        const int cd = c - (d ? a.outC[0] : 0);
        char* obase = (char*)(d ? a.out[1] : a.out[0]);
        const int oC = d ? a.outC[1] : a.outC[0], oacc = d ? a.out_acc[1] : a.out_acc[0];
        const int mz = mt / RG, my = (mt % RG) * TYM + j / TXM, mx = j % TXM;
        const int gz = z0 + mz, gy = y0 + my, gx = x0 + mx;
        if (wave < 4 && gz < a.oD && gy < a.oH && gx < a.oW && obase) {
            const size_t vox = ((size_t)gz * a.oH + gy) * a.oW + gx;
            uint2* p = (uint2*)(obase + (vox * oC + cd) * 2);
            float v0 = o4[0] + b4[0], v1 = o4[1] + b4[1], v2 = o4[2] + b4[2], v3 = o4[3] + b4[3];
            if (oacc) {
                const uint2 old = *p;
                v0 += bf_lo(old.x); v1 += bf_hi(old.x); v2 += bf_lo(old.y); v3 += bf_hi(old.y);
            }
            uint2 o;
            o.x = pack_bf16x2(v0, v1); o.y = pack_bf16x2(v2, v3);
            *p = o;
            const float r0 = bf_lo(o.x), r1 = bf_hi(o.x), r2 = bf_lo(o.y), r3 = bf_hi(o.y);
            if constexpr (BNS) {
                const uint2 ur = *(const uint2*)((const char*)a.bn_u + (vox * a.bn_C + c) * 2);
                const float uu[4] = {bf_lo(ur.x), bf_hi(ur.x), bf_lo(ur.y), bf_hi(ur.y)}, rr[4] = {r0, r1, r2, r3};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float mean = a.bn_stat[c + r], rstd = a.bn_stat[a.bn_C + c + r];
                    const float dv = rr[r] * act_d(fmaf(uu[r], a.bn_stat[2 * a.bn_C + c + r], a.bn_stat[3 * a.bn_C + c + r]), a.bn_act);
                    s1[r] = dv; s2[r] = dv * ((uu[r] - mean) * rstd);
                }
            } else {
                s1[0] = r0; s1[1] = r1; s1[2] = r2; s1[3] = r3;
                s2[0] = r0 * r0; s2[1] = r1 * r1; s2[2] = r2 * r2; s2[3] = r3 * r3;
            }
        }
    }
    float* const srows = BNS ? a.bn_partial : a.stats;
    if (srows) {   // statistics of the values as stored (BNS: the norm backward's): one partial row per tile
        __syncthreads();
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
            srows[((size_t)bid * g.Cout + nt0 * 16 + tid) * 2 + 0] = u;
            srows[((size_t)bid * g.Cout + nt0 * 16 + tid) * 2 + 1] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The network's first conv (Cin = 1, 3x3x3, stride 1, Cout = 16 * NT) on the matrix cores: K = the 27 taps padded to 32,
// D[co][voxel] = sum_tap W[co][tap] * x[voxel + tap].  The filter fragment is built once per block from the fp32 weights (no
// pack), the halo tile (6 x 10 x 18 bf16 = 2 KB) sits in LDS and a lane gathers its 8 taps with 2-byte LDS reads at
// compile-time offsets.  One MFMA per 16 voxels instead of 432 VALU FMAs per voxel (k_conv_first: 0.114 ms at 128^3, plus a
// separate statistics pass); the norm statistics come out of the epilogue as in k_mfma_conv_p.
// ------------------------------------------------------------------------------------------------
struct ConvFirstArgs {
    ConvGeom g;
    const void* x;       // bf16 [D][H][W] (Cin = 1), plain
    const float* w;      // fp32 [Cout][1][3][3][3]
    const float* bias;
    void* out;           // bf16 [D][H][W][Cout]
    float* stats;        // [gridDim.x][Cout][2] or nullptr
    int tiles_x, tiles_y, tiles_z;
};
template <int NT>
__global__ void __launch_bounds__(256, 2) k_conv_first_mfma(ConvFirstArgs a) {
    constexpr int BZ = 4, BY = 8, BX = 16, HZ = BZ + 2, HY = BY + 2, HX = BX + 2, NV = HZ * HY * HX;
    constexpr int ITERS = (NV + 255) / 256;
    __shared__ unsigned short tile[NV + 8];
    __shared__ float red[4 * NT * 16 * 2];
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, gq = lane >> 4;
    const int ntiles = a.tiles_x * a.tiles_y * a.tiles_z;
    const unsigned short* x = (const unsigned short*)a.x;

    // filter fragments: lane (row = co, k = 8*gq + e) <- w[co][tap = k], zero for the 5 padding taps
    bf16x8 wfrag[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int tap = 8 * gq + e;
            wfrag[n][e] = (__bf16)(tap < 27 ? a.w[(n * 16 + j) * 27 + tap] : 0.f);
        }
    // this lane's 8 tap addresses inside the tile for m-tile row 0 of its wave's z-plane (padding taps re-read tap 26: finite x 0)
    int abase[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int tap = 8 * gq + e < 27 ? 8 * gq + e : 26;
        abase[e] = ((((tap / 9) + wave) * HY + (tap / 3) % 3) * HX + tap % 3 + j) * 2;
    }
    float b4[NT][4];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) b4[n][r] = a.bias ? a.bias[n * 16 + gq * 4 + r] : 0.f;
    float s1[NT][4], s2[NT][4];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[n][r] = 0.f; s2[n][r] = 0.f; }

    // staging units: halo voxel u = tid + k*256
    int upk[ITERS];
    unsigned uvox[ITERS];
#pragma unroll
    for (int k = 0; k < ITERS; ++k) {
        const int u = tid + k * 256, hz = u / (HY * HX), hr = u % (HY * HX), hy = hr / HX, hx = hr % HX;
        upk[k] = u < NV ? (hz | (hy << 4) | (hx << 9)) : -1;
        uvox[k] = (unsigned)((hz * g.H + hy) * g.W + hx);
    }
    unsigned short R[ITERS];
    auto prefetch = [&](int t) {
        const int x0 = (t % a.tiles_x) * BX, y0 = ((t / a.tiles_x) % a.tiles_y) * BY, z0 = (t / (a.tiles_x * a.tiles_y)) * BZ;
        const int iz0 = z0 - 1, iy0 = y0 - 1, ix0 = x0 - 1;
        const long long org = ((long long)iz0 * g.H + iy0) * g.W + ix0;
#pragma unroll
        for (int k = 0; k < ITERS; ++k) {
            const int uc = upk[k];
            const int gz = iz0 + (uc & 15), gy = iy0 + ((uc >> 4) & 31), gx = ix0 + ((uc >> 9) & 63);
            R[k] = 0;
            if (uc >= 0 && (unsigned)gz < (unsigned)g.D && (unsigned)gy < (unsigned)g.H && (unsigned)gx < (unsigned)g.W)
                R[k] = x[org + uvox[k]];
        }
    };
    if ((int)blockIdx.x < ntiles) prefetch(blockIdx.x);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < ITERS; ++k)
            if (upk[k] >= 0) tile[tid + k * 256] = R[k];
        __syncthreads();
        if (t + (int)gridDim.x < ntiles) prefetch(t + gridDim.x);
        const int x0 = (t % a.tiles_x) * BX, y0 = ((t / a.tiles_x) % a.tiles_y) * BY, z0 = (t / (a.tiles_x * a.tiles_y)) * BZ;
        const int gz = z0 + wave, gx = x0 + j;
#pragma unroll
        for (int i = 0; i < BY; ++i) {          // m-tile = row i of z-plane `wave`: 16 voxels along x
            unsigned d[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned lo = *(const unsigned short*)((const char*)tile + abase[2 * e] + i * HX * 2);
                const unsigned hi = *(const unsigned short*)((const char*)tile + abase[2 * e + 1] + i * HX * 2);
                d[e] = lo | (hi << 16);
            }
            const bf16x8 xfrag = __builtin_bit_cast(bf16x8, make_uint4(d[0], d[1], d[2], d[3]));
            const int gy = y0 + i;
            const bool ok = gz < g.D && gy < g.H && gx < g.W;
            const size_t vox = ((size_t)gz * g.H + gy) * g.W + gx;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag[n], xfrag, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                if (ok) {
                    uint2 o;
                    o.x = pack_bf16x2(acc[0] + b4[n][0], acc[1] + b4[n][1]);
                    o.y = pack_bf16x2(acc[2] + b4[n][2], acc[3] + b4[n][3]);
                    *(uint2*)((char*)a.out + (vox * g.Cout + n * 16 + gq * 4) * 2) = o;
                    const float r0 = bf_lo(o.x), r1 = bf_hi(o.x), r2 = bf_lo(o.y), r3 = bf_hi(o.y);
                    s1[n][0] += r0; s1[n][1] += r1; s1[n][2] += r2; s1[n][3] += r3;
                    s2[n][0] = fmaf(r0, r0, s2[n][0]); s2[n][1] = fmaf(r1, r1, s2[n][1]);
                    s2[n][2] = fmaf(r2, r2, s2[n][2]); s2[n][3] = fmaf(r3, r3, s2[n][3]);
                }
            }
        }
    }
    if (a.stats) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float u = s1[n][r], v = s2[n][r];
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) { u += __shfl_xor(u, m); v += __shfl_xor(v, m); }
                if (j == 0) {
                    const int cl = n * 16 + gq * 4 + r;
                    red[(wave * NT * 16 + cl) * 2 + 0] = u;
                    red[(wave * NT * 16 + cl) * 2 + 1] = v;
                }
            }
        __syncthreads();
        if (tid < NT * 16) {
            float u = 0.f, v = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { u += red[(w * NT * 16 + tid) * 2]; v += red[(w * NT * 16 + tid) * 2 + 1]; }
            a.stats[((size_t)blockIdx.x * g.Cout + tid) * 2 + 0] = u;
            a.stats[((size_t)blockIdx.x * g.Cout + tid) * 2 + 1] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Sliding-window form for the single-chunk 3x3x3 stride-1 layers with 32 contraction channels (32->16 at 128^3, the 32->32
// layers at 64^3 and their dgrads).  k_mfma_conv_p stages a 6x6x18 halo (41 KB) per 256 output voxels: 2.5x more LDS writes and
// L2 reads than outputs, and LDS stores run at ~79 B/clk/CU -- as long as the tile's MFMAs.  Here a block owns an 8x16 (y,x)
// footprint and walks z: each step stages ONE 10x18 plane (11.5 KB for 128 outputs, 1.4x) into a ring of four planes while it
// computes the output plane whose three input planes are already resident -- one barrier per plane, the stores of plane p+1
// overlap the MFMAs of plane p-1.  Same packed filter, arguments, epilogue semantics and statistics rows (one per
// blockIdx.x) as k_mfma_conv_p<1,3,1,..,32,1,false>.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2) k_mfma_conv_z(MfmaConvArgs a, ZWork zw) {
    constexpr int BY = 8, BX = 16, HY = BY + 2, HX = BX + 2, HXP = 20, PLANE_B = HY * HXP * 64, RING = 4;
    constexpr int UNITS = HY * HX * 4, ITERS = (UNITS + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, gq = lane >> 4, lg = tid & 3;
    const int nt0 = blockIdx.y, NTT = g.Cout / 16, C0 = a.src[0].C;
    const bf16x8* wp = (const bf16x8*)a.w;

    // the (only) chunk's 27 filter fragments live in registers for the whole block (108 VGPRs; the plane ring needs few staging
    // registers, unlike the halo tiles of k_mfma_conv_p): the tap loop reads only patches from LDS, 1 KB per MFMA instead of 1.5
    bf16x8 wf[27];
#pragma unroll
    for (int ks = 0; ks < 27; ++ks) wf[ks] = wp[((size_t)ks * NTT + nt0) * 64 + lane];
    // Retire these 27 loads here, explicitly.  Otherwise the compiler cannot prove inside the plane loop that they have landed and
    // guards every use of wf[] with s_waitcnt vmcnt(n), n shrinking to 0 towards the last taps -- which also waits for the input
    // planes just requested for two steps later, i.e. it serialises the very prefetch the loop is built around.
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt/lgkmcnt untouched
    // this lane's patch addresses inside a plane for its two m-tiles (rows 2*wave, 2*wave + 1), per kx (swizzled as in k_mfma_conv_p)
    int mbase[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int hx = j + kx;
            mbase[i][kx] = ((2 * wave + i) * HXP + hx) * 64 + ((gq ^ (((hx >> 2) & 1) << 1)) << 4);
        }
    // staging units of a plane
    int ulds[ITERS], uyx[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int u = tid + it * 256, hv = u >> 2, hy = hv / HX, hx = hv % HX;
        ulds[it] = u < UNITS ? (hy * HXP + hx) * 64 + ((lg ^ (((hx >> 2) & 1) << 1)) << 4) : -1;
        uyx[it] = hy | (hx << 8);
    }
    const int c = lg * 8, sidx = (a.nsrc > 1 && c >= C0) ? 1 : 0;
    const char* sptr = (const char*)(sidx ? a.src[1].ptr : a.src[0].ptr) + (size_t)(c - (sidx ? C0 : 0)) * 2;
    const int sC = sidx ? a.src[1].C : C0;
    const unsigned vstride = (unsigned)sC * 2;

    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    float b4[4] = {0.f, 0.f, 0.f, 0.f};
    const int cch = nt0 * 16 + gq * 4;                         // this lane's 4 output channels
    if (a.bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r) b4[r] = a.bias[cch + r];
    }
    const int dsel = (a.nout > 1 && cch >= a.outC[0]) ? 1 : 0;
    const int cd = cch - (dsel ? a.outC[0] : 0);
    char* obase = (char*)(dsel ? a.out[1] : a.out[0]);
    const int oC = dsel ? a.outC[1] : a.outC[0], oacc = dsel ? a.out_acc[1] : a.out_acc[0];
    // Outputs leave through a buffer descriptor: a lane outside the volume stores at an offset beyond num_records and the hardware
    // drops it.  No branch around the store = a fixed number of memory operations per step, which is what lets the compiler wait
    // with vmcnt(n > 0) for the older input buffer while the younger one is still in flight (launch_conv_z checks the 2 GB bound).
    const unsigned obytes = obase ? (unsigned)((size_t)a.oD * a.oH * a.oW * oC * 2) : 0u;
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(obase, 0, (int)obytes, 0x00020000);
    constexpr int OOB = (int)0x80000000;

    const int nitems = zw.cols_x * zw.cols_y * zw.nseg;
    bf16x8 R[ITERS], R2[ITERS];
    const bf16x8 zero8 = __builtin_bit_cast(bf16x8, make_uint4(0u, 0u, 0u, 0u));
    // blocks of one XCD take a contiguous range of items, ordered segment-major: a compact patch of columns of one z segment whose
    // shared (y, x) halos are L2 hits (dealt round-robin every halo was fetched by another XCD; see k_mfma_conv_z16)
    const int ncols = zw.cols_x * zw.cols_y;
    for (int item = xcd_remap(blockIdx.x, gridDim.x); item < nitems; item += gridDim.x) {
        const int seg = item / ncols, col = item % ncols;
        const int x0 = (col % zw.cols_x) * BX, y0 = (col / zw.cols_x) * BY;
        const int zs = seg * zw.zlen, ze = zs + zw.zlen < g.D ? zs + zw.zlen : g.D;      // output planes [zs, ze)
        // per-column constants: which staging units lie inside the volume (y, x) and their address in plane 0 of the column
        unsigned umask = 0;
        const char* ubase[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int gy = y0 - 1 + (uyx[it] & 255), gx = x0 - 1 + (uyx[it] >> 8);
            const bool ok = ulds[it] >= 0 && (unsigned)gy < (unsigned)g.H && (unsigned)gx < (unsigned)g.W;
            if (ok) umask |= 1u << it;
            ubase[it] = sptr + (size_t)(ok ? gy * g.W + gx : 0) * vstride;
        }
        const size_t plane_bytes = (size_t)g.H * g.W * vstride;
        // The plane loads are issued as inline assembly and waited for by hand (wait_planes below).  Left to the compiler, every
        // use of a prefetched register is guarded by s_waitcnt vmcnt(0..2) -- it cannot count the younger operations across the
        // loop -- and a step then waits for the loads of the NEXT step's buffer as well, which is the latency this pipeline exists
        // to hide.  Loads are unconditional (units outside the volume read a valid dummy address and are zeroed when they are
        // stored to LDS), so every step issues exactly ITERS loads and, once it computes, exactly 2 buffer stores.
        auto prefetch = [&](int pz, bf16x8 (&Rr)[ITERS]) {
            const bool zin = (unsigned)pz < (unsigned)g.D;
            const size_t po = (size_t)(zin ? pz : 0) * plane_bytes;      // uniform
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const char* ad = ubase[it] + po;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(Rr[it]) : "v"(ad) : "memory");
            }
        };
        // vector-memory operations retire in issue order.  Younger than the buffer about to be stored are: the other buffer's ITERS
        // loads and, once steps compute, the 2 output stores of the previous step (those of the step before may also remain).
        // No VALU instruction may touch a prefetched register before this wait (the hardware does not interlock VGPR reads against
        // loads in flight): the registers go straight into ds_write (a memory operation, which the "memory" clobber keeps behind the
        // wait), out-of-volume units are zeroed by a second ds_write to the same address, and the wait has no register operands
        // (tied operands made the compiler copy the registers BEFORE the wait on one path).
        static_assert(ITERS == 3, "the vmcnt values below count 3 plane loads per step");
        auto wait_planes = [&](int younger_stores) {   // 0, 2 or 4 output stores are younger than the buffer, plus the other buffer's 3 loads
            if (younger_stores == 4) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            else if (younger_stores == 2) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        };
        // output pointers of this lane's two voxels, plane zs; advanced by one plane per step
        const int oy[2] = {y0 + 2 * wave, y0 + 2 * wave + 1}, ox = x0 + j;
        const bool ook[2] = {oy[0] < a.oH && ox < a.oW && obase != nullptr, oy[1] < a.oH && ox < a.oW && obase != nullptr};
        unsigned ooff[2];     // byte offsets inside the destination tensor (< 2^31)
#pragma unroll
        for (int i = 0; i < 2; ++i) ooff[i] = (unsigned)(((((size_t)zs * a.oH + (ook[i] ? oy[i] : 0)) * a.oW + (ook[i] ? ox : 0)) * oC + cd) * 2);
        const unsigned oplane = (unsigned)((size_t)a.oH * a.oW * oC * 2);

        // one output plane; SL = slot of input plane z-1 (compile-time: every LDS address is a per-lane base + an immediate)
        auto plane = [&](auto slc) {
            constexpr int SL = decltype(slc)::value;
            // A wave's two m-tiles are consecutive rows (2w, 2w+1): tap ky of row 2w+1 reads the same LDS patch as tap ky+1 of row
            // 2w.  Per (kz, kx) group the 4 distinct patch rows are read once and feed 6 MFMAs (2 rows x 3 ky): 36 LDS reads
            // per plane instead of 54, and 96 cycles of MFMA work behind every group of reads.
            // (two accumulation chains per m-tile -- four independent MFMA chains per wave -- measured no faster: 0.066 ms either way)
            f32x4 acc[2];
            bf16x8 xr[2][4];
            auto load_group = [&](int gi, int sl) {     // gi = kz*3 + kx
                const int kz = gi / 3, kx = gi % 3;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    xr[sl][r] = *(const bf16x8*)(smem + mbase[0][kx] + ((SL + kz) & 3) * PLANE_B + r * HXP * 64);
            };
            load_group(0, 0);
#pragma unroll
            for (int gi = 0; gi < 9; ++gi) {
                if (gi + 1 < 9) load_group(gi + 1, (gi + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
                const int kz = gi / 3, kx = gi % 3;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[(kz * 3 + ky) * 3 + kx], xr[gi & 1][i + ky],
                                                                         (gi == 0 && ky == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float v0 = acc[i][0] + b4[0], v1 = acc[i][1] + b4[1], v2 = acc[i][2] + b4[2], v3 = acc[i][3] + b4[3];
                if (oacc && ook[i]) {
                    const uint2 old = *(const uint2*)(obase + ooff[i]);
                    v0 += bf_lo(old.x); v1 += bf_hi(old.x); v2 += bf_lo(old.y); v3 += bf_hi(old.y);
                }
                typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
                u32x2 o;
                o.x = pack_bf16x2(v0, v1); o.y = pack_bf16x2(v2, v3);
                __builtin_amdgcn_raw_buffer_store_b64(o, orsrc, ook[i] ? (int)ooff[i] : OOB, 0, 0);
                if (a.stats && ook[i]) {
                    const float r0 = bf_lo(o.x), r1 = bf_hi(o.x), r2 = bf_lo(o.y), r3 = bf_hi(o.y);
                    s1[0] += r0; s1[1] += r1; s1[2] += r2; s1[3] += r3;
                    s2[0] = fmaf(r0, r0, s2[0]); s2[1] = fmaf(r1, r1, s2[1]); s2[2] = fmaf(r2, r2, s2[2]); s2[3] = fmaf(r3, r3, s2[3]);
                }
                ooff[i] += oplane;
            }
        };

        // Input planes are fetched TWO steps ahead: a step takes ~1.7 k cycles of MFMA work per SIMD, a global load under this
        // kernel's own traffic ~4-5 k, and with 216 VGPRs only two blocks share a CU -- fetched one step ahead, every step waited
        // for its loads.  Even planes travel through R, odd planes through R2 (the compiler's vmcnt then lets the younger
        // buffer's loads stay in flight while the older buffer is stored); the step is instantiated once per parity, each with
        // the two ring slots that parity can meet, so the code is no larger than the one-step-ahead form.
        // (An earlier two-steps-ahead attempt duplicated all four slot variants in both copies and was slower, 0.064 -> 0.076 ms.)
        // A step first computes output plane z = pz-2 -- its input planes pz-3..pz-1 were stored in earlier steps -- and only then
        // waits for plane pz, stores it and requests plane pz+2: the stores and their barrier have no consumer inside the step,
        // and a load has two steps plus one plane of MFMA work to arrive.
        auto step = [&](int pz, bf16x8 (&Rc)[ITERS], auto par) {
            constexpr int PAR = decltype(par)::value;      // parity of pz = parity of z
            const bool computes = pz >= zs + 2;
            if (computes) {   // slot(z-1) = z & 3, of parity PAR
                if (((pz - 2) & 2) == 0) plane(std::integral_constant<int, PAR>{});
                else plane(std::integral_constant<int, 2 + PAR>{});
            }
            const int slot = (pz + 1) & 3;
            const bool zin = (unsigned)pz < (unsigned)g.D;             // Rc holds plane pz
            wait_planes(pz >= zs + 3 ? 4 : (computes ? 2 : 0));       // stores of this step and of the previous one, if they computed
#pragma unroll
            for (int it = 0; it < ITERS; ++it)
                if (ulds[it] >= 0) {
                    char* d = smem + slot * PLANE_B + ulds[it];
                    const unsigned dl = (unsigned)(size_t)(__attribute__((address_space(3))) char*)d;    // LDS byte address
                    asm volatile("ds_write_b128 %0, %1" :: "v"(dl), "v"(Rc[it]) : "memory");
                    if (!(zin && ((umask >> it) & 1u))) *(bf16x8*)d = zero8;   // LDS operations of a wave execute in order
                }
            // always, for a fixed count of loads per step; past the segment's last input plane (ze) the load is repeated on plane ze,
            // which this block has just read (an L2 hit, never used) instead of fetching two more planes from HBM per segment
            prefetch(pz + 2 <= ze ? pz + 2 : ze, Rc);
            __syncthreads();
        };
        __syncthreads();                       // previous item's planes are no longer read
        // the previous item's last two steps prefetched past its end: let those loads land before their registers are reused
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // always start on an even plane (one idle step when zs - 1 is odd): R's loads are then older than R2's on every path
        const int pe0 = (zs - 1) & ~1;
        prefetch(pe0, R);
        prefetch(pe0 + 1, R2);
        for (int pe = pe0; pe <= ze + 1; pe += 2) {     // the last output plane ze-1 is computed at pz = ze+1
            step(pe, R, std::integral_constant<int, 0>{});
            if (pe + 1 <= ze + 1) step(pe + 1, R2, std::integral_constant<int, 1>{});
        }
    }
    // the last two steps prefetched past the end: those loads must land before the epilogue may reuse their registers
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a.stats) {
        float* red = (float*)smem;
        __syncthreads();
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
// sliding-window launcher: returns 0 if the geometry does not qualify, else gridDim.x
static int launch_conv_z(const MfmaConvArgs& a0, hipStream_t s) {
    const ConvGeom& g = a0.g;
    if (sliding_window_off() || g.Cin != 32 || g.Wo < 12 || g.Do < 8 || g.D != g.Do || g.H != g.Ho || g.W != g.Wo) return 0;
    for (int k = 0; k < 2; ++k)   // outputs are addressed with 31-bit byte offsets through a buffer descriptor
        if (a0.out[k] && (size_t)a0.oD * a0.oH * a0.oW * a0.outC[k] * 2 >= ((size_t)1 << 31)) return 0;
    ZWork zw;
    zw.cols_x = (g.Wo + 15) / 16; zw.cols_y = (g.Ho + 7) / 8;
    const int cols = zw.cols_x * zw.cols_y, gy = g.Cout / 16;
    int want = 512 / gy;                               // two blocks per CU in total
    if (want < 1) want = 1;
    int nseg = (want + cols - 1) / cols;               // z segments per column so that items >= the wanted block count
    if (nseg < 1) nseg = 1;
    int zlen = (g.Do + nseg - 1) / nseg;
    if (zlen < 4) zlen = 4;                            // at least 4 output planes per 2 warm-up planes
    nseg = (g.Do + zlen - 1) / zlen;
    zw.nseg = nseg; zw.zlen = zlen;
    const int items = cols * nseg;
    const int gx = items < want ? items : want;
    constexpr int lds = 4 * 10 * 20 * 64;
    static std::atomic<uint64_t> attr_done{0};
    set_max_lds_once(attr_done, (const void*)k_mfma_conv_z, lds);
    k_mfma_conv_z<<<dim3((unsigned)gx, (unsigned)gy), 256, lds, s>>>(a0, zw);
    return gx;
}

// ---- launch plumbing ----
template <int S, int KD, int PAD, int BZ, int BY, int BX, int CK, int NT, bool SC>
static int launch_cfg(const MfmaConvArgs& a0, hipStream_t s) {   // returns gridDim.x = the number of statistics partial rows
    // waves per block: 8 on the larger tiles when a wave then still owns >= 2 row tiles or the kind scatters (measured per variant:
    // 16->32 dgrad at 128^3 0.085 ms with 8 waves, 0.150 with 4); the one-row-tile kinds run 4 waves without the 128-VGPR cap
    constexpr int NW = ((BZ * BY * BX / 16) % 8 == 0 && BZ * BY * BX >= 256 && (NT > 1 || SC)) ? 8 : 4;
    MfmaConvArgs a = a0;
    a.tiles_x = (a.g.Wo + BX - 1) / BX; a.tiles_y = (a.g.Ho + BY - 1) / BY; a.tiles_z = (a.g.Do + BZ - 1) / BZ;
    constexpr bool SWZ = CK == 32 && S == 1 && KD == 3;
    constexpr int VS = SWZ ? 64 : (CK == 32 ? 96 : 32);
    constexpr int HXr = (BX - 1) * S + KD, HXP = SWZ ? (HXr + 3) / 4 * 4 : HXr;
    constexpr int T = KD * KD * KD, KSTEPS = CK == 32 ? T : (T + 1) / 2;
    constexpr size_t tile_b = (size_t)((BZ - 1) * S + KD) * ((BY - 1) * S + KD) * HXP * VS;
    constexpr size_t lds = tile_b + (KSTEPS * NT <= 32 ? (size_t)KSTEPS * NT * 1024 : 0);
    static_assert(tile_b >= NW * NT * 16 * 2 * 4, "stats scratch must fit the tile buffer");
    // two blocks per CU, except the stride-2 forward of the small levels with 32-channel chunks (16-128 blocks in all: one per CU)
    static_assert(lds <= 80 * 1024 || (S == 2 && KD == 3 && CK == 32 && lds <= 128 * 1024), "LDS per block");
    static std::atomic<uint64_t> attr_done{0};
    set_max_lds_once(attr_done, (const void*)k_mfma_conv_p<S, KD, PAD, BZ, BY, BX, CK, NT, SC, NW, false>, (int)lds);
    // persistent grid: at most ~2 blocks per CU in total (256 CUs), tiles strided over them
    const int tiles = a.tiles_x * a.tiles_y * a.tiles_z, gy = a.g.Cout / (16 * NT);
    // resident blocks per CU the LDS footprint allows, as a power of two (768-block grids measured 25 % slower than 512 / 1024)
    constexpr int bpc = (lds <= 40 * 1024 && NW == 4) ? 4 : 2;
    int gx = bpc * 256 / gy;
    if (gx < 1) gx = 1;
    if (gx > tiles) gx = tiles;
    dim3 grid((unsigned)gx, (unsigned)gy);
    if constexpr (NT == 1 && !SC) {
        // single-chunk layers (Cin == CK: 32->16 and 16->16 at 128^3): 4-wave blocks, <= 256 VGPRs, no spills
        // (measured against the 8-wave / 128-VGPR form: 0.087 -> 0.082 ms on the dominant layer, 4.16 -> 4.09 ms per step)
        if (a.g.Cin == CK) {
            static std::atomic<uint64_t> attr4_done{0};
            set_max_lds_once(attr4_done, (const void*)k_mfma_conv_p<S, KD, PAD, BZ, BY, BX, CK, NT, SC, 4, true>, (int)lds);
            k_mfma_conv_p<S, KD, PAD, BZ, BY, BX, CK, NT, SC, 4, true><<<grid, 256, lds, s>>>(a);
            return gx;
        }
    }
    k_mfma_conv_p<S, KD, PAD, BZ, BY, BX, CK, NT, SC, NW, false><<<grid, NW * 64, lds, s>>>(a);
    return gx;
}
template <int S, int KD, int PAD, int BZ, int BY, int BX, int CK, bool SC> static int launch_nt(const MfmaConvArgs& a, hipStream_t s) {
    int ntt = a.g.Cout / 16;
    if constexpr (SC) {   // scatter kinds always have ntt % 4 == 0 (rows = 8 * C, C % 16 == 0)
        // small volumes (the 4^3 / 8^3 levels: 1..8 tiles): fewer row tiles per block, so that more blocks stream the filter
        int tiles = ((a.g.Wo + BX - 1) / BX) * ((a.g.Ho + BY - 1) / BY) * ((a.g.Do + BZ - 1) / BZ);
        int nt = 4;
        if constexpr (BX <= 8) {
            while (nt > 1 && (int64_t)tiles * (ntt / nt) < 256) nt >>= 1;
            if (nt == 1) return launch_cfg<S, KD, PAD, BZ, BY, BX, CK, 1, true>(a, s);
            if (nt == 2) return launch_cfg<S, KD, PAD, BZ, BY, BX, CK, 2, true>(a, s);
        }
        return launch_cfg<S, KD, PAD, BZ, BY, BX, CK, 4, true>(a, s);
    } else {
        // row tiles per block: as many as divide the row count, fewer when the grid would not fill the 256 CUs
        int tiles = ((a.g.Wo + BX - 1) / BX) * ((a.g.Ho + BY - 1) / BY) * ((a.g.Do + BZ - 1) / BZ);
        int nt = ntt % 4 == 0 ? 4 : (ntt % 2 == 0 ? 2 : 1);
        constexpr int T = KD * KD * KD, KSTEPS = CK == 32 ? T : (T + 1) / 2;
        while (nt > 1 && KSTEPS * nt > 32) nt >>= 1;            // keep the chunk's filter fragments LDS-resident
        while (nt > 1 && (int64_t)tiles * (ntt / nt) < 256) nt >>= 1;
        // a single 16-channel chunk (the dgrad of decode0.0, 16 -> 32 at full resolution): two one-row-tile blocks, each staging the
        // filter slice once, beat one two-row-tile block
        if (CK == 16 && a.g.Cin == 16 && nt == 2 && S == 1 && KD == 3) nt = 1;
        if (nt == 4) return launch_cfg<S, KD, PAD, BZ, BY, BX, CK, 4, false>(a, s);
        if (nt == 2) return launch_cfg<S, KD, PAD, BZ, BY, BX, CK, 2, false>(a, s);
        return launch_cfg<S, KD, PAD, BZ, BY, BX, CK, 1, false>(a, s);
    }
}

// tile shapes per kind and output-grid width (every shape keeps the LDS tile <= 64 KB)
struct Tile { int bz, by, bx; };
static int tile_count(const ConvGeom& g, Tile t) { return ((g.Wo + t.bx - 1) / t.bx) * ((g.Ho + t.by - 1) / t.by) * ((g.Do + t.bz - 1) / t.bz); }
// small volumes (the deep levels) get small tiles so that tiles x row-tiles still covers the chip
static bool small_s1k3(const ConvGeom& g, int CK) {   // few big tiles: the volume is small, use k_mfma_conv_small (CK 32)
    Tile big = g.Wo >= 12 ? (CK == 32 ? Tile{4, 4, 16} : Tile{4, 8, 16}) : (g.Wo > 4 ? Tile{4, 8, 8} : Tile{4, 4, 4});
    return (int64_t)tile_count(g, big) * (g.Cout / 16) < 256;
}
static Tile tile_s1k3(const ConvGeom& g, int CK) {
    Tile big = g.Wo >= 12 ? (CK == 32 ? Tile{4, 4, 16} : Tile{4, 8, 16}) : (g.Wo > 4 ? Tile{4, 8, 8} : Tile{4, 4, 4});
    if (!small_s1k3(g, CK)) return big;
    if (CK == 32) return g.Wo > 4 ? Tile{2, 4, 8} : Tile{4, 4, 4};
    return g.Wo >= 12 ? Tile{2, 4, 16} : (g.Wo > 4 ? Tile{2, 4, 8} : Tile{4, 4, 4});
}
template <int BZ, int BY, int BX> static int launch_small(const MfmaConvArgs& a0, hipStream_t s) {
    MfmaConvArgs a = a0;
    a.tiles_x = (a.g.Wo + BX - 1) / BX; a.tiles_y = (a.g.Ho + BY - 1) / BY; a.tiles_z = (a.g.Do + BZ - 1) / BZ;
    constexpr size_t tile_b = (size_t)(BZ + 2) * (BY + 2) * ((BX + 2 + 3) / 4 * 4) * 64;
    static_assert(SMALL_QS * tile_b <= 160 * 1024 && tile_b >= 16 * 1024, "LDS budget / reduction scratch");
    const int nchunk = a.g.Cin / 32, nq = nchunk < SMALL_QS ? nchunk : SMALL_QS;
    const size_t lds = (size_t)nq * tile_b;
    static std::atomic<uint64_t> attr1_done{0}, attr2_done{0};
    set_max_lds_once(attr1_done, (const void*)k_mfma_conv_small<BZ, BY, BX, 1>, (int)(SMALL_QS * tile_b));
    set_max_lds_once(attr2_done, (const void*)k_mfma_conv_small<BZ, BY, BX, 2>, 80 * 1024);
    const int tiles = a.tiles_x * a.tiles_y * a.tiles_z;
    dim3 grid((unsigned)(a.g.Cout / 16), (unsigned)tiles);
    if (nchunk >= 8) {    // 256+ input channels (LDS above 80 KB: one block per CU anyway): eight waves, one chunk each per super-stage
        static std::atomic<uint64_t> attr5_done{0}, attr6_done{0};
        if (a.bn_partial) {
            set_max_lds_once(attr5_done, (const void*)k_mfma_conv_small<BZ, BY, BX, 1, true, 8>, (int)(SMALL_QS * tile_b));
            k_mfma_conv_small<BZ, BY, BX, 1, true, 8><<<grid, 512, lds, s>>>(a);
        } else {
            set_max_lds_once(attr6_done, (const void*)k_mfma_conv_small<BZ, BY, BX, 1, false, 8>, (int)(SMALL_QS * tile_b));
            k_mfma_conv_small<BZ, BY, BX, 1, false, 8><<<grid, 512, lds, s>>>(a);
        }
        return tiles;
    }
    if (a.bn_partial) {   // dgrad with the norm backward's statistics in the epilogue
        static std::atomic<uint64_t> attr3_done{0}, attr4_done{0};
        set_max_lds_once(attr3_done, (const void*)k_mfma_conv_small<BZ, BY, BX, 1, true>, (int)(SMALL_QS * tile_b));
        set_max_lds_once(attr4_done, (const void*)k_mfma_conv_small<BZ, BY, BX, 2, true>, 80 * 1024);
        if (lds <= 80 * 1024) k_mfma_conv_small<BZ, BY, BX, 2, true><<<grid, 256, lds, s>>>(a);
        else k_mfma_conv_small<BZ, BY, BX, 1, true><<<grid, 256, lds, s>>>(a);
        return tiles;
    }
    if (lds <= 80 * 1024) k_mfma_conv_small<BZ, BY, BX, 2><<<grid, 256, lds, s>>>(a);   // two blocks per CU
    else k_mfma_conv_small<BZ, BY, BX, 1><<<grid, 256, lds, s>>>(a);
    return tiles;
}
static int launch_s1k3(const MfmaConvArgs& a, int CK, hipStream_t s) {
    if (CK == 32) { int gz = launch_conv_z32(a, s); if (gz) return gz; gz = launch_conv_z(a, s); if (gz) return gz; }
    else { const int gz = launch_conv_z16(a, s); if (gz) return gz; }
    Tile t = tile_s1k3(a.g, CK);
    if (CK == 32 && small_s1k3(a.g, CK)) return t.bx == 8 ? launch_small<2, 4, 8>(a, s) : launch_small<4, 4, 4>(a, s);
    if (t.bx == 16 && t.bz == 4) { if (CK == 32) return launch_nt<1, 3, 1, 4, 4, 16, 32, false>(a, s); else return launch_nt<1, 3, 1, 4, 8, 16, 16, false>(a, s); }
    else if (t.bx == 16) { if (CK == 32) return launch_nt<1, 3, 1, 2, 4, 16, 32, false>(a, s); else return launch_nt<1, 3, 1, 2, 4, 16, 16, false>(a, s); }
    else if (t.bx == 8 && t.bz == 4) { if (CK == 32) return launch_nt<1, 3, 1, 4, 8, 8, 32, false>(a, s); else return launch_nt<1, 3, 1, 4, 8, 8, 16, false>(a, s); }
    else if (t.bx == 8) { if (CK == 32) return launch_nt<1, 3, 1, 2, 4, 8, 32, false>(a, s); else return launch_nt<1, 3, 1, 2, 4, 8, 16, false>(a, s); }
    else { if (CK == 32) return launch_nt<1, 3, 1, 4, 4, 4, 32, false>(a, s); else return launch_nt<1, 3, 1, 4, 4, 4, 16, false>(a, s); }
}
static Tile tile_s2k3(int Wo) { return Wo >= 12 ? Tile{2, 4, 16} : (Wo > 4 ? Tile{2, 4, 8} : Tile{4, 4, 4}); }
// stride-2 forward onto 8^3 voxels or fewer: 32-channel chunks (half as many pipeline stages; the 64-voxel tile's halo is 70 KB)
static bool s2_fwd_ck32(const ConvGeom& g) {
    return g.stride == 2 && g.ks == 3 && g.Cin % 32 == 0 && g.Wo <= 8;
}
static int launch_s2k3(const MfmaConvArgs& a, hipStream_t s) {   // CK 16 (halo of a stride-2 tile is 8x the output tile); small volumes: 32
    if (s2_fwd_ck32(a.g)) return a.g.Wo > 4 ? launch_nt<2, 3, 1, 2, 4, 8, 32, false>(a, s) : launch_nt<2, 3, 1, 4, 4, 4, 32, false>(a, s);
    if (a.g.Wo >= 12) return launch_nt<2, 3, 1, 2, 4, 16, 16, false>(a, s);
    else if (a.g.Wo > 4) return launch_nt<2, 3, 1, 2, 4, 8, 16, false>(a, s);
    else return launch_nt<2, 3, 1, 4, 4, 4, 16, false>(a, s);
}
// conv_trans dgrad onto 8^3 voxels or fewer: 32-channel chunks (half as many pipeline stages; the 64-voxel tile's halo is 49 KB)
static int convt_dgrad_ck(int cout_fwd, int coarseW) {
    return (cout_fwd % 32 == 0 && coarseW <= 8) ? 32 : 16;
}
static void launch_s2k2(const MfmaConvArgs& a, hipStream_t s) {   // conv_trans dgrad, CK 16 (small volumes: 32)
    if (convt_dgrad_ck(a.g.Cin, a.g.Wo) == 32) {
        if (a.g.Wo > 4) launch_nt<2, 2, 0, 2, 4, 8, 32, false>(a, s);
        else launch_nt<2, 2, 0, 4, 4, 4, 32, false>(a, s);
        return;
    }
    if (a.g.Wo >= 12) launch_nt<2, 2, 0, 2, 4, 16, 16, false>(a, s);
    else if (a.g.Wo > 4) launch_nt<2, 2, 0, 4, 8, 8, 16, false>(a, s);
    else launch_nt<2, 2, 0, 4, 4, 4, 16, false>(a, s);
}
static void launch_k1sc(const MfmaConvArgs& a, hipStream_t s) {   // conv_trans forward, CK 32
    if (a.g.Wo >= 12) launch_nt<1, 1, 0, 4, 8, 16, 32, true>(a, s);
    else if (a.g.Wo > 4) launch_nt<1, 1, 0, 4, 8, 8, 32, true>(a, s);
    else launch_nt<1, 1, 0, 4, 4, 4, 32, true>(a, s);
}
static void launch_k2sc(const MfmaConvArgs& a, hipStream_t s) {   // conv s2 dgrad, CK 32
    if (a.g.Wo >= 12) launch_nt<1, 2, 0, 4, 4, 16, 32, true>(a, s);
    else if (a.g.Wo > 4) launch_nt<1, 2, 0, 4, 8, 8, 32, true>(a, s);
    else launch_nt<1, 2, 0, 4, 4, 4, 32, true>(a, s);
}

static bool chan_ok(const ConvGeom& g, const SrcDesc* src, int nsrc) {
    if (g.Cin % 16 || g.Cout % 16) return false;
    for (int s = 0; s < nsrc; ++s)
        if (src[s].C % 16 || src[s].scale || src[s].act) return false;   // plain sources only (engine: activated copies)
    return true;
}
static MfmaConvArgs base_args() {
    MfmaConvArgs a;
    a.nsrc = 1; a.w = nullptr; a.bias = nullptr;
    a.out[0] = a.out[1] = nullptr; a.outC[0] = a.outC[1] = 0; a.out_acc[0] = a.out_acc[1] = 0; a.nout = 1;
    a.stats = nullptr; a.tiles_x = a.tiles_y = a.tiles_z = 0; a.sc_C = 0; a.oD = a.oH = a.oW = 0;
    return a;
}
static void set_dst(MfmaConvArgs& a, const DstGrad* dst, int ndst) {
    for (int k = 0; k < 2; ++k) {
        a.out[k] = k < ndst ? dst[k].ptr : nullptr;
        a.outC[k] = k < ndst ? dst[k].C : 0;
        a.out_acc[k] = k < ndst ? dst[k].accumulate : 0;
    }
    a.nout = ndst;
}

// ================= public: Conv3d 3x3x3, stride 1 or 2 =================
bool mfma_conv_fwd_supported(int dtype, const ConvGeom& g, const SrcDesc* src, int nsrc) {
    return dtype == 1 && g.ks == 3 && (g.stride == 1 || g.stride == 2) && chan_ok(g, src, nsrc);
}
static int fwd_ck(const ConvGeom& g) { return g.stride == 1 ? pick_ck(g.Cin, true) : (s2_fwd_ck32(g) ? 32 : 16); }
size_t mfma_conv_w_bytes(const ConvGeom& g) { return pack_bytes(g.Cin, g.Cout, fwd_ck(g), 27); }
size_t mfma_conv_dgrad_w_bytes(const ConvGeom& g) {
    if (g.stride == 1) return pack_bytes(g.Cout, g.Cin, pick_ck(g.Cout, true), 27);
    return pack_bytes(g.Cout, 8 * g.Cin, pick_ck(g.Cout, true), 8);
}
bool mfma_conv_dgrad_supported(int dtype, const ConvGeom& g, const SrcDesc* src, int nsrc) {
    if (!mfma_conv_fwd_supported(dtype, g, src, nsrc)) return false;
    return g.stride == 1 || g.Cout % 32 == 0;
}
void launch_mfma_pack_conv_w(const float* w, void* w_fwd, void* w_dgrad, const ConvGeom& g, hipStream_t s) {
    if (w_fwd) run_pack(w, w_fwd, g.Cin, g.Cout, fwd_ck(g), 27, PK_CONV_FWD, g.Cin, g.Cout, s);
    if (w_dgrad) {
        if (g.stride == 1) run_pack(w, w_dgrad, g.Cout, g.Cin, pick_ck(g.Cout, true), 27, PK_CONV_DGRAD, g.Cin, g.Cout, s);
        else run_pack(w, w_dgrad, g.Cout, 8 * g.Cin, 32, 8, PK_CONV_S2_DGRAD, g.Cin, g.Cout, s);
    }
}
// the same packs as launch_mfma_pack_conv_w, described for the batched pack kernel: out[0] forward, out[1] dgrad
int mfma_conv_pack_jobs(const ConvGeom& g, bool want_dgrad, PackJob* out) {
    out[0] = make_job(g.Cin, g.Cout, fwd_ck(g), 27, PK_CONV_FWD, g.Cin, g.Cout);
    if (!want_dgrad) return 1;
    if (g.stride == 1) out[1] = make_job(g.Cout, g.Cin, pick_ck(g.Cout, true), 27, PK_CONV_DGRAD, g.Cin, g.Cout);
    else out[1] = make_job(g.Cout, 8 * g.Cin, 32, 8, PK_CONV_S2_DGRAD, g.Cin, g.Cout);
    return 2;
}
int mfma_conv_blocks(const ConvGeom& g) {
    Tile t = g.stride == 1 ? tile_s1k3(g, fwd_ck(g)) : tile_s2k3(g.Wo);
    return tile_count(g, t);
}
int launch_mfma_conv_fwd(const ConvGeom& g, const SrcDesc* src, int nsrc, const void* w_mfma, const float* bias, void* out,
                         float* stats_partial, hipStream_t s) {
    MfmaConvArgs a = base_args();
    a.g = g; a.nsrc = nsrc; a.src[0] = src[0]; if (nsrc > 1) a.src[1] = src[1];
    a.w = w_mfma; a.bias = bias;
    a.out[0] = out; a.outC[0] = g.Cout;
    a.stats = stats_partial;
    a.oD = g.Do; a.oH = g.Ho; a.oW = g.Wo;
    if (g.stride == 2 && fwd_ck(g) == 16) {      // the sliding-window form reads the 16-channel-chunk pack
        const int rows = launch_s2_conv_fwd(g, src, nsrc, w_mfma, bias, out, stats_partial, s);
        if (rows) return rows;
    }
    return g.stride == 1 ? launch_s1k3(a, fwd_ck(g), s) : launch_s2k3(a, s);
}
// dgrad (g = forward geometry).  stride 1: 27-tap conv of dL/dy with the flipped filter.  stride 2: 8-tap conv of
// dL/dy on the coarse grid producing all 8 output parities at once (rows = 8*Cin), scattered to 2*m + parity.
int launch_mfma_conv_dgrad(const ConvGeom& g, const void* dy, const void* w_mfma_dgrad, const DstGrad* dst, int ndst, hipStream_t s,
                           const BnBwdStats* bn) {
    MfmaConvArgs a = base_args();
    a.src[0].ptr = dy; a.src[0].C = g.Cout;
    a.w = w_mfma_dgrad;
    set_dst(a, dst, ndst);
    a.g.Cin = g.Cout; a.g.D = g.Do; a.g.H = g.Ho; a.g.W = g.Wo; a.g.ks = 3; a.g.stride = 1;
    a.oD = g.D; a.oH = g.H; a.oW = g.W;
    if (g.stride == 1) {
        a.g.Cout = g.Cin; a.g.Do = g.D; a.g.Ho = g.H; a.g.Wo = g.W;
        static const bool no_bn = getenv("UNET_NO_DGRAD_BNSTATS") != nullptr;
        if (bn && !no_bn && ndst == 1 && dst[0].ptr && !dst[0].accumulate && bn->C == g.Cin && pick_ck(g.Cout, true) == 16 && conv_z16_applies(a)) {
            a.bn_u = bn->u; a.bn_stat = bn->stat; a.bn_partial = bn->partial; a.bn_act = bn->act; a.bn_C = bn->C;
            return launch_conv_z16(a, s);
        }
        // ... and of k_mfma_conv_small (the 16^3 and smaller levels; the sliding-window kernels need Cin == 32 and never take these)
        static const bool no_bn_small = getenv("UNET_NO_DGRAD_BNSTATS_SMALL") != nullptr;
        if (bn && !no_bn && !no_bn_small && ndst == 1 && dst[0].ptr && !dst[0].accumulate && bn->C == g.Cin && pick_ck(g.Cout, true) == 32 &&
            a.g.Cin != 32 && small_s1k3(a.g, 32)) {
            a.bn_u = bn->u; a.bn_stat = bn->stat; a.bn_partial = bn->partial; a.bn_act = bn->act; a.bn_C = bn->C;
            return launch_s1k3(a, 32, s);
        }
        launch_s1k3(a, pick_ck(g.Cout, true), s);
    } else {
        static const bool no_bn2 = getenv("UNET_NO_DGRAD_BNSTATS") != nullptr;
        const int rows = launch_s2_conv_dgrad(g, dy, w_mfma_dgrad, dst, ndst, s, no_bn2 ? nullptr : bn);
        if (rows) return rows > 0 ? rows : 0;
        a.g.Cout = 8 * g.Cin; a.sc_C = g.Cin;
        a.g.Do = (g.D + 1) / 2; a.g.Ho = (g.H + 1) / 2; a.g.Wo = (g.W + 1) / 2;   // coarse positions m with 2m or 2m+1 inside the volume
        launch_k2sc(a, s);
    }
    return 0;
}

// ================= public: ConvTranspose3d 2x2x2 stride 2 =================
bool mfma_convt_supported(int dtype, const ConvGeom& g, const SrcDesc* src, int nsrc) {
    return dtype == 1 && chan_ok(g, src, nsrc) && g.Cin % 32 == 0;
}
size_t mfma_convt_w_bytes(const ConvGeom& g) { return pack_bytes(g.Cin, 8 * g.Cout, 32, 1); }
size_t mfma_convt_dgrad_w_bytes(const ConvGeom& g) { return pack_bytes(g.Cout, g.Cin, convt_dgrad_ck(g.Cout, g.W), 8); }
void launch_mfma_pack_convt_w(const float* w, void* w_fwd, void* w_dgrad, const ConvGeom& g, hipStream_t s) {
    if (w_fwd) run_pack(w, w_fwd, g.Cin, 8 * g.Cout, 32, 1, PK_CONVT_FWD, g.Cin, g.Cout, s);
    if (w_dgrad) run_pack(w, w_dgrad, g.Cout, g.Cin, convt_dgrad_ck(g.Cout, g.W), 8, PK_CONVT_DGRAD, g.Cin, g.Cout, s);
}
int mfma_convt_pack_jobs(const ConvGeom& g, PackJob* out) {
    out[0] = make_job(g.Cin, 8 * g.Cout, 32, 1, PK_CONVT_FWD, g.Cin, g.Cout);
    out[1] = make_job(g.Cout, g.Cin, convt_dgrad_ck(g.Cout, g.W), 8, PK_CONVT_DGRAD, g.Cin, g.Cout);
    return 2;
}
void launch_mfma_convt_fwd(const ConvGeom& g, const SrcDesc* src, int nsrc, const void* w_mfma, const float* bias, void* out, hipStream_t s) {
    if (launch_s2_convt_fwd(g, src, nsrc, w_mfma, bias, out, s)) return;
    MfmaConvArgs a = base_args();
    a.g = g; a.g.Cout = 8 * g.Cout; a.g.Do = g.D; a.g.Ho = g.H; a.g.Wo = g.W; a.g.ks = 1; a.g.stride = 1;
    a.nsrc = nsrc; a.src[0] = src[0]; if (nsrc > 1) a.src[1] = src[1];
    a.w = w_mfma; a.bias = bias;
    a.out[0] = out; a.outC[0] = g.Cout;
    a.sc_C = g.Cout; a.oD = g.Do; a.oH = g.Ho; a.oW = g.Wo;
    launch_k1sc(a, s);
}
void launch_mfma_convt_dgrad(const ConvGeom& g, const void* dy, const void* w_mfma_dgrad, const DstGrad* dst, int ndst, hipStream_t s) {
    if (convt_dgrad_ck(g.Cout, g.W) == 16 && launch_s2_convt_dgrad(g, dy, w_mfma_dgrad, dst, ndst, s)) return;
    MfmaConvArgs a = base_args();
    a.src[0].ptr = dy; a.src[0].C = g.Cout;
    a.w = w_mfma_dgrad;
    set_dst(a, dst, ndst);
    a.g.Cin = g.Cout; a.g.Cout = g.Cin; a.g.D = g.Do; a.g.H = g.Ho; a.g.W = g.Wo; a.g.Do = g.D; a.g.Ho = g.H; a.g.Wo = g.W;
    a.g.ks = 2; a.g.stride = 2;
    a.oD = g.D; a.oH = g.H; a.oW = g.W;
    launch_s2k2(a, s);
}

// ================= public: the first conv (Cin = 1) =================
bool conv_first_mfma_supported(int dtype, const ConvGeom& g, const SrcDesc* src, int nsrc) {
    return dtype == 1 && nsrc == 1 && g.Cin == 1 && g.ks == 3 && g.stride == 1 && (g.Cout == 16 || g.Cout == 32) && !src[0].scale &&
           src[0].act == 0 && (int64_t)8 * g.H * g.W < (1ll << 31);
}
int conv_first_mfma_blocks(const ConvGeom& g) {
    int tiles = ((g.W + 15) / 16) * ((g.H + 7) / 8) * ((g.D + 3) / 4);
    // 73 VGPRs (Cout 16) allow 6 blocks per CU; the kernel is a per-wave latency chain (LDS gathers -> MFMA -> store), so the grid, not
    // the registers, set its occupancy at the round-2 value of 512
    constexpr int want = 512;
    return tiles < want ? tiles : want;
}
int launch_conv_first_mfma(const ConvGeom& g, const SrcDesc* src, const float* w, const float* bias, void* out, float* stats_partial,
                           hipStream_t s) {
    ConvFirstArgs a;
    a.g = g; a.x = src[0].ptr; a.w = w; a.bias = bias; a.out = out; a.stats = stats_partial;
    a.tiles_x = (g.W + 15) / 16; a.tiles_y = (g.H + 7) / 8; a.tiles_z = (g.D + 3) / 4;
    const int nb = conv_first_mfma_blocks(g);
    if (g.Cout == 16) k_conv_first_mfma<1><<<nb, 256, 0, s>>>(a);
    else k_conv_first_mfma<2><<<nb, 256, 0, s>>>(a);
    return nb;
}

}  // namespace unet
