// MFMA weight-gradient kernels for gfx950 (bf16 storage, fp32 accumulate).
//
//   conv 3x3x3 (stride 1 or 2):   dW[co][ci][t] = sum_v  a[v*S + t - 1][ci] * dy[v][co]      (a = transformed conv input)
//   conv_trans 2x2x2 stride 2:    dW[ci][co][t] = sum_v  x[v][ci] * dy[2v + t][co]           (x = transformed input)
// Both are T implicit GEMMs  D_t[ca][cb] += A_t[ca][k] * B[k][cb]  with k = voxel of the tile-side ("B") tensor and the
// halo-side ("A") tensor read at k*S + t - PAD:  conv: A = input, B = dy;  conv_trans: A = dy (fine grid), B = input.
//   * K runs over voxels, but both tensors are channels-last, so both MFMA operands are read with the
//     hardware-transposing ds_read_b64_tr_b16 (4 voxels x 16 channels per 16-lane group) from LDS planes
//     [16-channel tile][voxel][16 ch] (32-B voxels: a 32-lane half covers one whole 256-B bank row).
//   * one K-step = 32 tile voxels = 8 groups of 4 consecutive x; lane group gq and read r fetch group gq + 4r of
//     the step, identically for A and B.
//   * a block = PI x PJ (ca-tile, cb-tile) pairs, one pair per wave (4/(PI*PJ) waves per pair split the K-steps);
//     every wave keeps all T tap accumulators in registers (27 taps: 108 VGPRs) across the block's tiles.
//   * no float atomics: each block writes its partial to a slab [split][Cb][Ca][T] (the gradient's layout); a second kernel
//     sums the slabs in a fixed order and adds into the fp32 gradient (+=, as .grad accumulates).
//   * conv: dL/dbias = sum_v dy[v][co] is accumulated by the threads that stage dy (blocks of ca-tile 0 only).
#include <type_traits>
#include "mfma_util.h"

namespace unet {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

struct MfmaWgradArgs {
    ConvGeom g;        // Cin = Ca (halo-side channels), Cout = Cb (tile-side channels); D,H,W = A volume; Do,Ho,Wo = B volume
    SrcDesc asrc[2];   // halo side (may be a channel concat)
    int nasrc;
    SrcDesc bsrc;      // tile side
    float* slab;       // [nsplit][Cb][Ca][T]
    float* bias_slab;  // [nsplit][Cb] or nullptr: per-channel sums of the RAW tile-side tensor
    int tiles_x, tiles_y, tiles_z;
    // direct != 0 (one block per (ca, cb) pair walks ALL tiles: gridDim.x == 1): `slab` IS the gradient tensor (same layout) and
    // `bias_slab` the bias gradient; the block ADDS its result to them (.grad accumulates) -- no slab, no reduce pass.  Every
    // gradient element belongs to exactly one block and is summed in a fixed order, so the result stays bit-reproducible.
    int direct;
    int bias_from_a;   // conv_trans: bias_slab holds per-channel sums of the HALO-side tensor (dy), [nsplit][Ca]
};

__device__ __forceinline__ bf16x8 tr_read2(const char* p0, const char* p1) {
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p1);
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int S, int KD, int PAD, int BZ, int BY, int BX, int PI, int PJ>
__global__ void __launch_bounds__(256, 2) k_mfma_wgrad(MfmaWgradArgs a) {   // <= 256 VGPRs: two blocks per CU
    constexpr int HZ = (BZ - 1) * S + KD, HY = (BY - 1) * S + KD, HX = (BX - 1) * S + KD;
    constexpr int NVA = HZ * HY * HX, NVB = BZ * BY * BX, T = KD * KD * KD;
    constexpr int PLANE_A = NVA * 32, PLANE_B = NVB * 32, B_OFF = PI * PLANE_A;
    constexpr int P = PI * PJ, WPP = 4 / P;       // pairs per block, waves per pair
    constexpr int GPR = BX / 4;                   // 4-voxel groups per row
    constexpr int KSTEPS = NVB / 32;
    static_assert(P == 1 || P == 2 || P == 4, "pairs per block");
    static_assert(NVB % 32 == 0 && BX % 4 == 0, "tile must hold whole K-steps");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, il = lane & 15, gq = lane >> 4;
    const int q4 = il >> 2, p4 = il & 3;
    const int pw = wave % P, kw = wave / P;
    const int it_ = pw / PJ, jt = pw % PJ;
    const int COTB = (g.Cout / 16) / PJ;
    const int ciB = (blockIdx.y / COTB) * PI, coB = (blockIdx.y % COTB) * PJ;
    const int ntiles = a.tiles_x * a.tiles_y * a.tiles_z;
    const int C0 = a.asrc[0].C;

    // TS (one pair per block): the 4 waves split the TAPS (wave w owns taps w, w+4, ...) instead of the K-steps: 7 accumulators
    // instead of 27 leave registers for a 4-deep ring of A fragments -- with 27 accumulators only one tr-read pair could be in
    // flight per MFMA and the tap loop ran at LDS latency (~100 cycles per 16-cycle MFMA at 2 waves per SIMD).
    constexpr bool TS = P == 1;
    constexpr int TW = (T + 3) / 4, NACC = TS ? TW : T;
    f32x4 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum[e] = 0.f;

    // staging roles (fixed per thread): A units (plane, half) and B units (plane, half).  Sources are plain (a tensor with a
    // pending norm/activation is read through its activated copy).
    constexpr int GA = PI * 2, GB = PJ * 2;
    const int ua = tid % GA, ub = tid % GB;
    const int ca = (ciB + (ua >> 1)) * 16 + (ua & 1) * 8;
    const int sa = (a.nasrc > 1 && ca >= C0) ? 1 : 0;
    const int cla = ca - (sa ? C0 : 0);
    const int aC = sa ? a.asrc[1].C : C0;
    const char* abase = (const char*)(sa ? a.asrc[1].ptr : a.asrc[0].ptr) + (size_t)cla * 2;
    const int cb = (coB + (ub >> 1)) * 16 + (ub & 1) * 8;
    const char* bbase = (const char*)a.bsrc.ptr + (size_t)cb * 2;
    const bool do_bias = a.bias_slab != nullptr && ciB == 0 && !a.bias_from_a;
    // conv_trans: dL/dbias = sum over the FINE grid of dy, which is this kernel's halo side; with a 2x2x2 stride-2 window the halo
    // tiles cover the fine grid exactly once, so the blocks of cb-tile 0 sum the A units they stage (threads with equal tid % GA hold
    // the same 8 channels, as the B side does for conv)
    const bool do_bias_a = a.bias_slab != nullptr && coB == 0 && a.bias_from_a;

    // staging units of this thread, decoded once: coordinates (z 4 bits | y 5 << 4 | x 6 << 9) | LDS offset / 16 << 15 (or -1),
    // and the voxel offset from the tile's origin in the source volume (address = tile base + offset * voxel stride)
    constexpr int UNITS_A = NVA * GA, ITERS_A = (UNITS_A + 255) / 256;
    constexpr int UNITS_B = NVB * GB, ITERS_B = (UNITS_B + 255) / 256;
    static_assert(HZ <= 16 && HY <= 32 && HX <= 64, "staging unit packing");
    int ua_pk[ITERS_A], ub_pk[ITERS_B];
    unsigned ua_vox[ITERS_A], ub_vox[ITERS_B];
#pragma unroll
    for (int itr = 0; itr < ITERS_A; ++itr) {
        const int u = tid + itr * 256, hv = u / GA;
        const int hz = hv / (HY * HX), hr = hv % (HY * HX), hy = hr / HX, hx = hr % HX;
        const int lds_off = (ua >> 1) * PLANE_A + hv * 32 + (ua & 1) * 16;
        ua_pk[itr] = u < UNITS_A ? (hz | (hy << 4) | (hx << 9) | ((lds_off >> 4) << 15)) : -1;
        ua_vox[itr] = (unsigned)((hz * g.H + hy) * g.W + hx);
    }
#pragma unroll
    for (int itr = 0; itr < ITERS_B; ++itr) {
        const int u = tid + itr * 256, tv = u / GB;
        const int tz = tv / (BY * BX), tr = tv % (BY * BX), ty = tr / BX, tx = tr % BX;
        const int lds_off = B_OFF + (ub >> 1) * PLANE_B + tv * 32 + (ub & 1) * 16;
        ub_pk[itr] = u < UNITS_B ? (tz | (ty << 4) | (tx << 9) | ((lds_off >> 4) << 15)) : -1;
        ub_vox[itr] = (unsigned)((tz * g.Ho + ty) * g.Wo + tx);
    }
    // Prefetch registers are native vectors (HIP's uint4 struct copies become memcpy's that stay in scratch memory).
    bf16x8 RA[ITERS_A], RB[ITERS_B];
    const bf16x8 zero8 = __builtin_bit_cast(bf16x8, make_uint4(0u, 0u, 0u, 0u));
    auto prefetch = [&](int tile, auto direct) {   // direct: store every unit to LDS as it arrives (no prefetch registers)
        constexpr bool DIRECT = decltype(direct)::value;
        const int ox0 = (tile % a.tiles_x) * BX, oy0 = ((tile / a.tiles_x) % a.tiles_y) * BY, oz0 = (tile / (a.tiles_x * a.tiles_y)) * BZ;
        const int ix0 = ox0 * S - PAD, iy0 = oy0 * S - PAD, iz0 = oz0 * S - PAD;
        const char* at = abase + ((((long long)iz0 * g.H + iy0) * g.W + ix0) * aC) * 2;
        const char* bt = bbase + ((((long long)oz0 * g.Ho + oy0) * g.Wo + ox0) * g.Cout) * 2;
        const unsigned avs = (unsigned)aC * 2, bvs = (unsigned)g.Cout * 2;
        const bool ain = iz0 >= 0 && iy0 >= 0 && ix0 >= 0 && iz0 + HZ <= g.D && iy0 + HY <= g.H && ix0 + HX <= g.W;
        const bool bin = oz0 + BZ <= g.Do && oy0 + BY <= g.Ho && ox0 + BX <= g.Wo;
#pragma unroll
        for (int itr = 0; itr < ITERS_A; ++itr) {
            const int uc = ua_pk[itr];
            bool ok = (itr + 1) * 256 <= UNITS_A || uc >= 0;
            if (!ain) {
                const int gz = iz0 + (uc & 15), gy = iy0 + ((uc >> 4) & 31), gx = ix0 + ((uc >> 9) & 63);
                ok = ok && (unsigned)gz < (unsigned)g.D && (unsigned)gy < (unsigned)g.H && (unsigned)gx < (unsigned)g.W;
            }
            bf16x8 v = zero8;
            if (ok) v = *(const bf16x8*)(at + __umul24(ua_vox[itr], avs));
            if constexpr (DIRECT) {
                if ((itr + 1) * 256 <= UNITS_A || uc >= 0) {
                    *(bf16x8*)(smem + ((uc >> 15) << 4)) = v;
                    if (do_bias_a) {
                        const uint4 w = __builtin_bit_cast(uint4, v);
                        bsum[0] += bf_lo(w.x); bsum[1] += bf_hi(w.x); bsum[2] += bf_lo(w.y); bsum[3] += bf_hi(w.y);
                        bsum[4] += bf_lo(w.z); bsum[5] += bf_hi(w.z); bsum[6] += bf_lo(w.w); bsum[7] += bf_hi(w.w);
                    }
                }
            }
            else RA[itr] = v;
        }
#pragma unroll
        for (int itr = 0; itr < ITERS_B; ++itr) {
            const int uc = ub_pk[itr];
            bool ok = (itr + 1) * 256 <= UNITS_B || uc >= 0;
            if (!bin) {
                const int gz = oz0 + (uc & 15), gy = oy0 + ((uc >> 4) & 31), gx = ox0 + ((uc >> 9) & 63);
                ok = ok && gz < g.Do && gy < g.Ho && gx < g.Wo;
            }
            bf16x8 v = zero8;
            if (ok) v = *(const bf16x8*)(bt + __umul24(ub_vox[itr], bvs));
            if constexpr (DIRECT) {
                if ((itr + 1) * 256 <= UNITS_B || uc >= 0) {
                    *(bf16x8*)(smem + ((uc >> 15) << 4)) = v;
                    if (do_bias) {
                        const uint4 w = __builtin_bit_cast(uint4, v);
                        bsum[0] += bf_lo(w.x); bsum[1] += bf_hi(w.x); bsum[2] += bf_lo(w.y); bsum[3] += bf_hi(w.y);
                        bsum[4] += bf_lo(w.z); bsum[5] += bf_hi(w.z); bsum[6] += bf_lo(w.w); bsum[7] += bf_hi(w.w);
                    }
                }
            } else RB[itr] = v;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int itr = 0; itr < ITERS_A; ++itr)
            if ((itr + 1) * 256 <= UNITS_A || ua_pk[itr] >= 0) {
                *(bf16x8*)(smem + ((ua_pk[itr] >> 15) << 4)) = RA[itr];
                if (do_bias_a) {
                    const uint4 v = __builtin_bit_cast(uint4, RA[itr]);
                    bsum[0] += bf_lo(v.x); bsum[1] += bf_hi(v.x); bsum[2] += bf_lo(v.y); bsum[3] += bf_hi(v.y);
                    bsum[4] += bf_lo(v.z); bsum[5] += bf_hi(v.z); bsum[6] += bf_lo(v.w); bsum[7] += bf_hi(v.w);
                }
            }
#pragma unroll
        for (int itr = 0; itr < ITERS_B; ++itr) {
            if ((itr + 1) * 256 <= UNITS_B || ub_pk[itr] >= 0) {
                *(bf16x8*)(smem + ((ub_pk[itr] >> 15) << 4)) = RB[itr];
                if (do_bias) {
                    const uint4 v = __builtin_bit_cast(uint4, RB[itr]);
                    bsum[0] += bf_lo(v.x); bsum[1] += bf_hi(v.x); bsum[2] += bf_lo(v.y); bsum[3] += bf_hi(v.y);
                    bsum[4] += bf_lo(v.z); bsum[5] += bf_hi(v.z); bsum[6] += bf_lo(v.w); bsum[7] += bf_hi(v.w);
                }
            }
        }
    };

    const char* pa = smem + it_ * PLANE_A + p4 * 8;
    const char* pb = smem + B_OFF + jt * PLANE_B + p4 * 8;
    // PREF: the next tile's global loads ride in registers through the MFMA phase (27 tap accumulators = 108 VGPRs leave room
    // for ~10 staging units per thread); the wider configurations load and store back to back as before.
    constexpr bool PREF = KD == 2 || (S == 1 && PI == 1 && ITERS_A + ITERS_B <= 8);
    // tap-split MFMA phase of wave W: iterations n = (K-step s, own tap i) fully unrolled, every LDS address = a per-lane base
    // + a compile-time offset, RD A-fragment read pairs in flight
    auto ts_phase = [&](auto wc) {
        constexpr int W = decltype(wc)::value;
        constexpr int NTW = (T - W + 3) / 4, NIT = KSTEPS * NTW, RD = 4;
        const int lr = gq / GPR, lxg = gq % GPR;
        const char* pal = pa + ((lr * S) * HX + (4 * lxg + q4) * S) * 32;
        const char* pbl = pb + (lr * BX + 4 * lxg + q4) * 32;
        auto aoff = [](int s, int r, int t) {
            const int cr = (8 * s + 4 * r) / GPR, z = cr / BY, y = cr % BY;
            return ((z * S * HY + y * S) * HX) * 32 + (((t / (KD * KD)) * HY + (t / KD) % KD) * HX + t % KD) * 32;
        };
        auto boff = [](int s, int r) {
            const int cr = (8 * s + 4 * r) / GPR, z = cr / BY, y = cr % BY;
            return ((z * BY + y) * BX) * 32;
        };
        bf16x8 ring[RD], bfr[2];
        bfr[0] = tr_read2(pbl + boff(0, 0), pbl + boff(0, 1));
#pragma unroll
        for (int n = 0; n < RD; ++n)
            if (n < NIT) ring[n] = tr_read2(pal + aoff(n / NTW, 0, W + 4 * (n % NTW)), pal + aoff(n / NTW, 1, W + 4 * (n % NTW)));
#pragma unroll
        for (int n = 0; n < NIT; ++n) {
            const int s = n / NTW, i = n % NTW;
            if (i == 0 && s + 1 < KSTEPS) bfr[(s + 1) & 1] = tr_read2(pbl + boff(s + 1, 0), pbl + boff(s + 1, 1));
            __builtin_amdgcn_sched_barrier(0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[n % RD], bfr[s & 1], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (n + RD < NIT) {
                const int m = n + RD;
                ring[m % RD] = tr_read2(pal + aoff(m / NTW, 0, W + 4 * (m % NTW)), pal + aoff(m / NTW, 1, W + 4 * (m % NTW)));
            }
        }
    };
    if (PREF && (int)blockIdx.x < ntiles) prefetch(blockIdx.x, std::false_type{});
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __syncthreads();                 // every wave is done reading the previous tile
        if constexpr (PREF) commit(); else prefetch(tile, std::true_type{});
        __syncthreads();
        if (PREF && tile + (int)gridDim.x < ntiles) prefetch(tile + gridDim.x, std::false_type{});   // in flight during the MFMAs below
        if constexpr (TS) {
            switch (wave) {
                case 0: ts_phase(std::integral_constant<int, 0>{}); break;
                case 1: ts_phase(std::integral_constant<int, 1>{}); break;
                case 2: ts_phase(std::integral_constant<int, 2>{}); break;
                default: ts_phase(std::integral_constant<int, 3>{}); break;
            }
            continue;
        }
        // ---- MFMA: this wave's K-steps; the tr-reads of tap t+1 are issued before the MFMA of tap t ----
        if constexpr (!TS) {
#pragma unroll 1
        for (int s = kw; s < KSTEPS; s += WPP) {
            int ao[2], bo[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                int grp = 8 * s + 4 * r + gq;
                int row = grp / GPR, xg = grp % GPR;
                int z = row / BY, y = row % BY, x = 4 * xg + q4;
                ao[r] = ((z * S * HY + y * S) * HX + x * S) * 32;
                bo[r] = ((z * BY + y) * BX + x) * 32;
            }
            const bf16x8 bfrag = tr_read2(pb + bo[0], pb + bo[1]);
            bf16x8 abuf[2];
            abuf[0] = tr_read2(pa + ao[0], pa + ao[1]);
#pragma unroll
            for (int t = 0; t < T; ++t) {
                if (t + 1 < T) {
                    const int t1 = t + 1, toff = (((t1 / (KD * KD)) * HY + (t1 / KD) % KD) * HX + t1 % KD) * 32;
                    abuf[t1 & 1] = tr_read2(pa + ao[0] + toff, pa + ao[1] + toff);
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(abuf[t & 1], bfrag, acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        }
    }

    // Slab tile of a (ca, cb) pair in the gradient's layout [cb][ca][t]: a cb row is 16*T contiguous floats.  A lane owns cb = il and
    // ca = gq*4 .. +3 (16-B pieces 4*T*4 B apart): stored from the registers, every wave instruction touched 64 different lines.  The
    // tile goes through LDS (row pitch 16*T + 4 floats: at most 2-way conflicts) and leaves as whole rows.
    constexpr int RP = 16 * T + 4, ROW4 = 4 * T;      // row pitch (floats), float4 per row
    if constexpr (TS) {
        // every wave holds finished taps (wave w: taps w, w+4, ...)
        __syncthreads();
        float* stg = (float*)smem;
#pragma unroll
        for (int i = 0; i < TW; ++i)
            if (wave + 4 * i < T) {
#pragma unroll
                for (int r = 0; r < 4; ++r) stg[il * RP + (gq * 4 + r) * T + wave + 4 * i] = acc[i][r];
            }
        __syncthreads();
        float* base = a.slab + (size_t)blockIdx.x * T * g.Cin * g.Cout + ((size_t)coB * 16 * g.Cin + (size_t)ciB * 16) * T;
        if (a.direct) {
            // the gradient itself: read-modify-write.  A flat parameter buffer only guarantees 4-B alignment of a tensor (6-float head
            // biases may sit in front of it): 16-B accesses when the tensor happens to be aligned, scalars otherwise.
            if ((reinterpret_cast<uintptr_t>(base) & 15) == 0) {
                for (int q = tid; q < 16 * ROW4; q += 256) {
                    const int row = q / ROW4, c4 = q % ROW4;
                    f32x4* d = (f32x4*)(base + (size_t)row * g.Cin * T + c4 * 4);
                    const f32x4 o = *d, v = *(const f32x4*)(stg + row * RP + c4 * 4);
                    *d = f32x4{o[0] + v[0], o[1] + v[1], o[2] + v[2], o[3] + v[3]};
                }
            } else {
                for (int q = tid; q < 16 * 16 * T; q += 256) {
                    const int row = q / (16 * T), c = q % (16 * T);
                    base[(size_t)row * g.Cin * T + c] += stg[row * RP + c];
                }
            }
        } else {
        for (int q = tid; q < 16 * ROW4; q += 256) {
            const int row = q / ROW4, c4 = q % ROW4;
            *(f32x4*)(base + (size_t)row * g.Cin * T + c4 * 4) = *(const f32x4*)(stg + row * RP + c4 * 4);
        }
        }
    }
    if constexpr (!TS) {
    // ---- reduce the K-split waves of each pair through LDS, then write the slab ----
    __syncthreads();
    float* red = (float*)smem;   // [P][T][64][4]
#pragma unroll 1
    for (int kk = 1; kk < WPP; ++kk) {
        if (kw == kk) {
#pragma unroll
            for (int t = 0; t < T; ++t) *(f32x4*)(red + ((pw * T + t) * 64 + lane) * 4) = acc[t];
        }
        __syncthreads();
        if (kw == 0) {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                f32x4 o = *(const f32x4*)(red + ((pw * T + t) * 64 + lane) * 4);
                acc[t][0] += o[0]; acc[t][1] += o[1]; acc[t][2] += o[2]; acc[t][3] += o[3];
            }
        }
        __syncthreads();
    }
    // The slab has the gradient's own layout [cb][ca][t] (torch: [Cout][Cin][k^3], conv_trans [Cin][Cout][8]), so the reduce
    // kernel is a linear, coalesced sum; pair by pair the tile is transposed through LDS and stored as whole rows (see above).
    {
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
            float* base = a.slab + (size_t)blockIdx.x * T * g.Cin * g.Cout +
                          ((size_t)(coB + pr % PJ) * 16 * g.Cin + (size_t)(ciB + pr / PJ) * 16) * T;
            for (int q = tid; q < 16 * ROW4; q += 256) {
                const int row = q / ROW4, c4 = q % ROW4;
                *(f32x4*)(base + (size_t)row * g.Cin * T + c4 * 4) = *(const f32x4*)(stg + row * RP + c4 * 4);
            }
        }
    }
    }
    if (do_bias || do_bias_a) {
        // threads with equal tid % GU hold partial sums of the same 8 channels (GU = the side's staging roles): shuffle tree inside each
        // wave (lanes GU apart), then the four wave totals through LDS (the serial loop over 256 / GU LDS values per output cost ~6 us
        // at the end of every block)
        const int GU = do_bias_a ? GA : GB;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = bsum[e];
            for (int m = GU; m < 64; m <<= 1) v += __shfl_xor(v, m);
            bsum[e] = v;
        }
        __syncthreads();
        float* bred = (float*)smem;   // [4 waves][GU][8]
        if (lane < GU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) bred[(wave * GU + lane) * 8 + e] = bsum[e];
        }
        __syncthreads();
        if (tid < GU * 8) {
            int u = tid / 8, e = tid % 8;
            float sacc = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) sacc += bred[(w * GU + u) * 8 + e];
            const int nC = do_bias_a ? g.Cin : g.Cout;      // channels of the summed side (conv_trans: the kernel's Cin = the layer's Cout)
            int c = ((do_bias_a ? ciB : coB) + (u >> 1)) * 16 + (u & 1) * 8 + e;
            if (a.direct) a.bias_slab[c] += sacc;     // the bias gradient itself (exactly one block per channel)
            else a.bias_slab[(size_t)blockIdx.x * nC + c] = sacc;
        }
    }
}

// dw[i] += sum_split slab[split][i]  (slab and dw share the layout);  db[cb] += sum_split bias_slab[split][cb].
// A block = LX quads of 4 consecutive outputs x LY split lanes: every lane streams 16-B slab reads (8 in flight), the split
// lanes are summed through LDS in a fixed order (bit-reproducible).  n and Cb are multiples of 4 (16-channel tiles); dw / db are
// only 4-B aligned (a flat parameter buffer has 6-float head biases in front of later tensors), so they are updated by scalars.
// (The first version gave each thread ONE float and nsplit/8 dependent 4-B loads: 43 us for a 14 MB slab = 0.33 TB/s.)
template <int LX, int LY>
__global__ void __launch_bounds__(256) k_mfma_wgrad_reduce(const float* __restrict__ slab, const float* __restrict__ bias_slab, int nsplit,
                                                           int64_t n, int Cb, float* __restrict__ dw, float* __restrict__ db) {
    static_assert(LX * LY == 256, "block shape");
    __shared__ double red[LY > 1 ? LY : 1][LX][4];
    const int64_t ntot = n + (db && bias_slab ? Cb : 0);
    const int lx = threadIdx.x % LX, ly = threadIdx.x / LX;
    const int64_t i = ((int64_t)blockIdx.x * LX + lx) * 4;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (i < ntot) {
        const float* base = i < n ? slab + i : bias_slab + (i - n);
        const int64_t stride = i < n ? n : Cb;
        int k = ly;
        for (; k + 7 * LY < nsplit; k += 8 * LY) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *(const float4*)(base + (int64_t)(k + u * LY) * stride);
#pragma unroll
            for (int u = 0; u < 8; ++u) { s0 += v[u].x; s1 += v[u].y; s2 += v[u].z; s3 += v[u].w; }
        }
        for (; k < nsplit; k += LY) {
            const float4 v = *(const float4*)(base + (int64_t)k * stride);
            s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
        }
    }
    if (LY > 1) {
        red[ly][lx][0] = s0; red[ly][lx][1] = s1; red[ly][lx][2] = s2; red[ly][lx][3] = s3;
        __syncthreads();
        if (ly == 0) {
            s0 = s1 = s2 = s3 = 0.0;
#pragma unroll
            for (int k = 0; k < LY; ++k) { s0 += red[k][lx][0]; s1 += red[k][lx][1]; s2 += red[k][lx][2]; s3 += red[k][lx][3]; }
        }
    }
    if (ly == 0 && i < ntot) {
        float* d = i < n ? dw + i : db + (i - n);
        d[0] += (float)s0; d[1] += (float)s1; d[2] += (float)s2; d[3] += (float)s3;
    }
}
void wgrad_reduce(const float* slab, const float* bias_slab, int nsplit, int64_t n, int Cb, float* dw, float* db, hipStream_t s) {
    const int64_t ntot = n + (db && bias_slab ? Cb : 0), quads = (ntot + 3) / 4;
    if (nsplit <= 8) k_mfma_wgrad_reduce<256, 1><<<cdiv64(quads, 256), 256, 0, s>>>(slab, bias_slab, nsplit, n, Cb, dw, db);
    else if (nsplit <= 64) k_mfma_wgrad_reduce<64, 4><<<cdiv64(quads, 64), 256, 0, s>>>(slab, bias_slab, nsplit, n, Cb, dw, db);
    else k_mfma_wgrad_reduce<16, 16><<<cdiv64(quads, 16), 256, 0, s>>>(slab, bias_slab, nsplit, n, Cb, dw, db);
}

// The same sum for MANY layers in one launch (a plan's sliding-window wgrads leave their slabs in the workspace; the backward
// sums them once per gradient bucket instead of once per layer: ~20 dependent launches fewer per step, and the small layers'
// reduces fill the chip together).  Offsets are in floats from the workspace base / the flat gradient buffer.
__global__ void __launch_bounds__(256) k_wgrad_reduce_batched(const WgradReduceJob* __restrict__ jobs, int job0, int njobs, int blk_base,
                                                              const float* __restrict__ ws, float* __restrict__ gflat) {
    __shared__ double red[256][4];
    __shared__ int sj;
    if (threadIdx.x == 0) {
        const int b = (int)blockIdx.x + blk_base;
        int j = job0;
        for (int k = job0; k < job0 + njobs; ++k)
            if (b >= jobs[k].blk0) j = k;
        sj = j;
    }
    __syncthreads();
    const WgradReduceJob jb = jobs[sj];
    const int LY = jb.ly, LX = 256 / LY;
    const int lx = threadIdx.x % LX, ly = threadIdx.x / LX;
    const bool hb = jb.bias_off >= 0;
    const long long n = jb.n, ntot = n + (hb ? jb.Cb : 0);
    const long long i = ((long long)((int)blockIdx.x + blk_base - jb.blk0) * LX + lx) * 4;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (i < ntot) {
        const float* base = i < n ? ws + jb.slab_off + i : ws + jb.bias_off + (i - n);
        const long long stride = i < n ? n : jb.Cb;
        int k = ly;
        for (; k + 7 * LY < jb.nsplit; k += 8 * LY) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *(const float4*)(base + (long long)(k + u * LY) * stride);
#pragma unroll
            for (int u = 0; u < 8; ++u) { s0 += v[u].x; s1 += v[u].y; s2 += v[u].z; s3 += v[u].w; }
        }
        for (; k < jb.nsplit; k += LY) {
            const float4 v = *(const float4*)(base + (long long)k * stride);
            s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
        }
    }
    if (LY > 1) {
        red[threadIdx.x][0] = s0; red[threadIdx.x][1] = s1; red[threadIdx.x][2] = s2; red[threadIdx.x][3] = s3;
        __syncthreads();
        if (ly == 0) {
            s0 = s1 = s2 = s3 = 0.0;
            for (int k = 0; k < LY; ++k) { s0 += red[k * LX + lx][0]; s1 += red[k * LX + lx][1]; s2 += red[k * LX + lx][2]; s3 += red[k * LX + lx][3]; }
        }
    }
    if (ly == 0 && i < ntot) {
        float* d = i < n ? gflat + jb.dw_off + i : gflat + jb.db_off + (i - n);
        d[0] += (float)s0; d[1] += (float)s1; d[2] += (float)s2; d[3] += (float)s3;
    }
}
// fills ly / blk0 / nblk of a job (host side, plan creation); returns its block count
int wgrad_reduce_job_blocks(WgradReduceJob& j, int blk0) {
    j.ly = j.nsplit > 64 ? 16 : (j.nsplit > 8 ? 4 : 1);
    const long long quads = (j.n + (j.bias_off >= 0 ? j.Cb : 0) + 3) / 4;
    const int lx = 256 / j.ly;
    j.blk0 = blk0;
    j.nblk = (int)((quads + lx - 1) / lx);
    return j.nblk;
}
void launch_wgrad_reduce_batched(const WgradReduceJob* jobs_dev, int job0, int njobs, int blk_base, int nblocks, const void* ws, float* gflat,
                                 hipStream_t s) {
    if (njobs <= 0 || nblocks <= 0) return;
    k_wgrad_reduce_batched<<<(unsigned)nblocks, 256, 0, s>>>(jobs_dev, job0, njobs, blk_base, (const float*)ws, gflat);
}

// ------------------------------------------------------------------------------------------------
// wgrad of the network's first conv (Cin = 1, 3x3x3 stride 1, Cout = 16 * NT) on the matrix cores:
//     D[tap][co] = sum_voxels x[voxel + tap] * dy[voxel][co]        M = 27 taps (+ row 27 = ones: the bias gradient), K = voxels
// A tile is 2 x 8 rows of 32 voxels; the input halo sits in LDS three times, shifted by 0 / 1 / 2 elements, so that the 8
// consecutive voxels a lane needs for tap kx are one aligned 16-B read; dy is gathered with 2-byte reads (voxel stride padded
// against bank conflicts).  4 waves split the rows; partial D's are summed through LDS and written as one slab row per block.
// (The VALU kernel it replaces, k_wgrad_first, was the exposed tail of the backward: it needs dL/d(raw) of the first norm layer.)
// ------------------------------------------------------------------------------------------------
struct FirstWgradArgs {
    ConvGeom g;
    const void* x;      // bf16 [D][H][W]
    const void* dy;     // bf16 [D][H][W][Cout]
    float* slab;        // [gridDim.x][27 * Cout]
    float* bslab;       // [gridDim.x][Cout]
    int tiles_x, tiles_y, tiles_z;
    // FUSE: dy is dL/d(activated view) of the conv's output and the norm backward's element-wise pass is applied as the tile is staged
    // (k_norm_bwd_apply8's arithmetic, rounded to bf16 exactly as that pass stores it: the same bits reach the MFMAs) -- the first conv's
    // dL/d(raw output) has no other reader, so the pass (read 2, write 1 tensor of 64 MB at 128^3) is never run.
    const void* u;      // raw conv output, bf16 [D][H][W][Cout]
    const float* stat;  // [4][Cout] mean, rstd, scale, shift
    const float* coef;  // [3][Cout] k_norm_bwd_finalize's coefficients
    int act;
};
template <int NT, bool FUSE>
__global__ void __launch_bounds__(256, 2) k_wgrad_first_mfma(FirstWgradArgs a) {
    constexpr int TZ = 2, TY = 8, TX = 32, HZ = TZ + 2, HY = TY + 2, NROW = TZ * TY, CO = 16 * NT;
    constexpr int XCOPY = HZ * HY * TX;                 // elements of one shifted copy of the halo
    constexpr int VSB = CO * 2 + 4;                     // bytes per dy voxel in LDS (padded: k-groups 8 voxels apart hit different banks)
    constexpr int DY_OFF = 3 * XCOPY * 2;
    __shared__ __attribute__((aligned(16))) char sm[DY_OFF + NROW * TX * VSB + 16];
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, gq = lane >> 4;
    const unsigned short* x = (const unsigned short*)a.x;
    const char* dyb = (const char*)a.dy;
    // A operand: lane (row = tap, k group gq); taps 27..31 of the second row tile read tap 26's patch (finite; rows never stored)
    int abase[2];
    bool ones_row = false;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        int tap = rt * 16 + j;
        if (tap == 27) ones_row = true;
        if (tap > 26) tap = 26;
        const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
        abase[rt] = (kx * XCOPY + (kz * HY + ky) * TX + gq * 8) * 2;
    }
    const bf16x8 ones = __builtin_bit_cast(bf16x8, make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u));
    f32x4 acc[2][NT];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[rt][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ntiles = a.tiles_x * a.tiles_y * a.tiles_z;
    // A tile's global data (halo elements and dy units of this thread) rides in registers while the previous tile is computed: the
    // loop used to be load -> LDS -> barrier -> compute per tile, 8 tiles per block, every load latency exposed (45 us at 128^3 for
    // 71 MB) -- and this kernel is the LAST of the backward, on the critical path of the step.
    constexpr int XI = (HZ * HY * (TX + 2) + 255) / 256, DI = NROW * TX * (CO / 8) / 256;
    static_assert(NROW * TX * (CO / 8) % 256 == 0, "dy units per thread");
    unsigned short xr[XI];
    uint4 dr[DI], ur[FUSE ? DI : 1];
    unsigned dmask = 0;
    // a thread's dy units all lie in the same 8-channel group (256 % (CO / 8) == 0): its coefficients stay in registers
    float cA[8], cB[8], cD[8], csc[8], csh[8];
    if constexpr (FUSE) {
        const int c0 = (tid % (CO / 8)) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c0 + e;
            const float mean = a.stat[c], rstd = a.stat[CO + c];
            csc[e] = a.stat[2 * CO + c]; csh[e] = a.stat[3 * CO + c];
            cA[e] = a.coef[c];
            cB[e] = -a.coef[c] * rstd * a.coef[2 * CO + c];
            cD[e] = -a.coef[c] * (a.coef[CO + c] - mean * rstd * a.coef[2 * CO + c]);
        }
    }
    const char* ub = (const char*)a.u;
    auto gload = [&](int t) {
        const int x0 = (t % a.tiles_x) * TX, y0 = ((t / a.tiles_x) % a.tiles_y) * TY, z0 = (t / (a.tiles_x * a.tiles_y)) * TZ;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int e = tid + i * 256;
            const int hx = e % (TX + 2), r = e / (TX + 2), hy = r % HY, hz = r / HY;
            const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            unsigned short v = 0;
            if (e < HZ * HY * (TX + 2) && (unsigned)gz < (unsigned)g.D && (unsigned)gy < (unsigned)g.H && (unsigned)gx < (unsigned)g.W)
                v = x[((size_t)gz * g.H + gy) * g.W + gx];
            xr[i] = v;
        }
#pragma unroll
        for (int i = 0; i < DI; ++i) {
            const int u = tid + i * 256;
            const int c8 = u % (CO / 8), vx = (u / (CO / 8)) % TX, row = u / ((CO / 8) * TX);
            const int gz = z0 + row / TY, gy = y0 + row % TY, gx = x0 + vx;
            uint4 v = make_uint4(0u, 0u, 0u, 0u), w = make_uint4(0u, 0u, 0u, 0u);
            const bool in = gz < g.D && gy < g.H && gx < g.W;
            const size_t off = ((((size_t)gz * g.H + gy) * g.W + gx) * CO + c8 * 8) * 2;
            if (in) v = *(const uint4*)(dyb + off);
            if constexpr (FUSE) {
                if (in) w = *(const uint4*)(ub + off);
                ur[i] = w;
                dmask = in ? (dmask | (1u << i)) : (dmask & ~(1u << i));   // units outside the volume keep dy = 0
            }
            dr[i] = v;
        }
    };
    auto lstore = [&]() {
        // input halo, three shifted copies: copy s holds x[.., x0 - 1 + i + s] at element i
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int e = tid + i * 256;
            if (e < HZ * HY * (TX + 2)) {
                const int hx = e % (TX + 2), r = e / (TX + 2), hy = r % HY, hz = r / HY;
#pragma unroll
                for (int sft = 0; sft < 3; ++sft) {
                    const int k = hx - sft;
                    if (k >= 0 && k < TX) *(unsigned short*)(sm + (sft * XCOPY + (hz * HY + hy) * TX + k) * 2) = xr[i];
                }
            }
        }
        // dy tile [row][voxel][co] (4-B LDS stores)
#pragma unroll
        for (int i = 0; i < DI; ++i) {
            const int u = tid + i * 256;
            const int c8 = u % (CO / 8), vx = (u / (CO / 8)) % TX, row = u / ((CO / 8) * TX);
            unsigned* d = (unsigned*)(sm + DY_OFF + (row * TX + vx) * VSB + c8 * 16);
            if constexpr (FUSE) {
                if ((dmask >> i) & 1u) {
                    const unsigned gu[4] = {dr[i].x, dr[i].y, dr[i].z, dr[i].w}, uu[4] = {ur[i].x, ur[i].y, ur[i].z, ur[i].w};
                    unsigned o[4];
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const float g0 = __uint_as_float(gu[e / 2] << 16), g1 = __uint_as_float(gu[e / 2] & 0xffff0000u);
                        const float u0 = __uint_as_float(uu[e / 2] << 16), u1 = __uint_as_float(uu[e / 2] & 0xffff0000u);
                        const float r0 = fmaf(cA[e] * g0, act_d(fmaf(u0, csc[e], csh[e]), a.act), fmaf(cB[e], u0, cD[e]));
                        const float r1 = fmaf(cA[e + 1] * g1, act_d(fmaf(u1, csc[e + 1], csh[e + 1]), a.act), fmaf(cB[e + 1], u1, cD[e + 1]));
                        o[e / 2] = (unsigned)__bfloat16_as_ushort(__float2bfloat16(r0)) | ((unsigned)__bfloat16_as_ushort(__float2bfloat16(r1)) << 16);
                    }
                    dr[i] = make_uint4(o[0], o[1], o[2], o[3]);
                }
            }
            d[0] = dr[i].x; d[1] = dr[i].y; d[2] = dr[i].z; d[3] = dr[i].w;
        }
    };
    if ((int)blockIdx.x < ntiles) gload(blockIdx.x);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        __syncthreads();               // the previous tile's LDS image is no longer read
        lstore();
        __syncthreads();
        if (t + (int)gridDim.x < ntiles) gload(t + gridDim.x);
#pragma unroll 1
        for (int row = wave; row < NROW; row += 4) {      // one K-step = the 32 voxels of output row `row`
            const int rz = row / TY, ry = row % TY;
            const int roff = (rz * HY + ry) * TX * 2;
            bf16x8 af[2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) af[rt] = *(const bf16x8*)(sm + abase[rt] + roff);
            if (ones_row) af[1] = ones;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                unsigned d[4];
                const char* pb = sm + DY_OFF + (row * TX + gq * 8) * VSB + (n * 16 + j) * 2;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned lo = *(const unsigned short*)(pb + (2 * e) * VSB), hi = *(const unsigned short*)(pb + (2 * e + 1) * VSB);
                    d[e] = lo | (hi << 16);
                }
                const bf16x8 bf = __builtin_bit_cast(bf16x8, make_uint4(d[0], d[1], d[2], d[3]));
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) acc[rt][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[rt], bf, acc[rt][n], 0, 0, 0);
            }
        }
    }
    // sum the four waves' partial D through LDS, write the slab row: dw[co][0][tap] then db[co]
    __syncthreads();
    float* red = (float*)sm;                                // [4 waves][2][NT][64 lanes][4]
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int n = 0; n < NT; ++n) *(f32x4*)(red + (((wave * 2 + rt) * NT + n) * 64 + lane) * 4) = acc[rt][n];
    __syncthreads();
    const int O = 27 * g.Cout;
    float* sl = a.slab + (size_t)blockIdx.x * O;
    for (int e = tid; e < 2 * NT * 256; e += 256) {        // e = ((rt*NT + n)*64 + lane')*4 + reg
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) v += red[w * 2 * NT * 256 + e];
        const int reg = e & 3, l2 = (e >> 2) & 63, rn = e >> 8, n = rn % NT, rt = rn / NT;
        const int tap = rt * 16 + 4 * (l2 >> 4) + reg, co = n * 16 + (l2 & 15);
        if (tap < 27) sl[co * 27 + tap] = v;
        else if (tap == 27) a.bslab[(size_t)blockIdx.x * g.Cout + co] = v;
    }
}
bool conv_first_wgrad_mfma_supported(int dtype, const ConvGeom& g, const SrcDesc* src, int nsrc) {
    return dtype == 1 && nsrc == 1 && g.Cin == 1 && g.ks == 3 && g.stride == 1 && (g.Cout == 16 || g.Cout == 32) && !src[0].scale &&
           src[0].act == 0;
}
static int first_wgrad_blocks(const ConvGeom& g) {
    int tiles = ((g.W + 31) / 32) * ((g.H + 7) / 8) * ((g.D + 1) / 2);
    return tiles < 512 ? tiles : 512;
}
size_t conv_first_wgrad_mfma_scratch_bytes(const ConvGeom& g) { return (size_t)first_wgrad_blocks(g) * (27 * g.Cout + g.Cout) * 4 + 256; }
// dw += , db += (db may be null); scratch: conv_first_wgrad_mfma_scratch_bytes
int conv_first_wgrad_splits(const ConvGeom& g) { return first_wgrad_blocks(g); }
void launch_conv_first_wgrad_mfma(const ConvGeom& g, const SrcDesc* src, const void* dy, float* dw, float* db, void* scratch, hipStream_t s,
                                  bool defer_reduce, const NormBwdFuse* nb_fuse) {
    FirstWgradArgs a;
    a.g = g; a.x = src[0].ptr; a.dy = dy; a.slab = (float*)scratch;
    a.u = nb_fuse ? nb_fuse->u : nullptr; a.stat = nb_fuse ? nb_fuse->stat : nullptr; a.coef = nb_fuse ? nb_fuse->coef : nullptr;
    a.act = nb_fuse ? nb_fuse->act : 0;
    a.tiles_x = (g.W + 31) / 32; a.tiles_y = (g.H + 7) / 8; a.tiles_z = (g.D + 1) / 2;
    const int nb = first_wgrad_blocks(g);
    const int64_t O = 27 * g.Cout;
    a.bslab = a.slab + (size_t)nb * O;
    if (nb_fuse) {
        if (g.Cout == 16) k_wgrad_first_mfma<1, true><<<nb, 256, 0, s>>>(a);
        else k_wgrad_first_mfma<2, true><<<nb, 256, 0, s>>>(a);
    } else {
        if (g.Cout == 16) k_wgrad_first_mfma<1, false><<<nb, 256, 0, s>>>(a);
        else k_wgrad_first_mfma<2, false><<<nb, 256, 0, s>>>(a);
    }
    if (!defer_reduce) wgrad_reduce(a.slab, db ? a.bslab : nullptr, nb, O, g.Cout, dw, db, s);
}

static bool chan16(const ConvGeom& g, const SrcDesc* src, int nsrc) {
    if (g.Cin % 16 || g.Cout % 16) return false;
    for (int s = 0; s < nsrc; ++s)
        if (src[s].C % 16 || src[s].scale || src[s].act) return false;   // plain sources only (engine: activated copies)
    return true;
}
bool mfma_wgrad_supported(int dtype, const ConvGeom& g, const SrcDesc* src, int nsrc) {
    return dtype == 1 && g.ks == 3 && (g.stride == 1 || g.stride == 2) && chan16(g, src, nsrc);
}
bool mfma_convt_wgrad_supported(int dtype, const ConvGeom& g, const SrcDesc* src, int nsrc) {
    return dtype == 1 && nsrc == 1 && chan16(g, src, nsrc);
}

// kind 0: conv stride 1, 1: conv stride 2, 2: conv_trans.  Ca/Cb and the tile-side grid width decide the configuration.
struct WgradCfg { int bz, by, bx, pi, pj, nsplit, gy, direct, polite = 0; };
// Small volumes (tile side 4^3 or 8^3 voxels: the 8^3 / 4^3 levels of the default architecture, K = 64..512 voxels, gradients of
// 2..14 MB): output-stationary.  One block per (ca, cb) pair owns the pair's T tap tiles for the WHOLE volume (4 waves split the
// taps), walks the volume's tiles and adds its result straight into the gradient: >= 128 blocks, no slab write, no reduce read.
// (Round 2 ran these as 64..128 blocks of 2x2 pairs with one LDS read pair in flight per MFMA, each writing a 110-KB slab tile:
// 20..43 us per layer for < 0.3 % of the step's FLOPs.)
static bool wgrad_direct_cfg(int kind, int bD, int bH, int bW, WgradCfg& c) {
    if (bD != bH || bH != bW) return false;
    if (bW == 4) { c.bz = 4; c.by = 4; c.bx = 4; }
    else if (bW == 8 && kind == 0) { c.bz = 8; c.by = 8; c.bx = 8; }      // stride 1: the 10^3 halo of a 16-channel tile is 32 KB
    else if (bW == 8) { c.bz = 2; c.by = 8; c.bx = 8; }                   // stride 2 / conv_trans: halo planes of 17^2 / 16^2 voxels, 4 tiles along z
    else return false;
    c.pi = c.pj = 1; c.nsplit = 1; c.direct = 1;
    return true;
}
static WgradCfg wgrad_cfg(int kind, int Ca, int Cb, int bD, int bH, int bW) {
    WgradCfg c;
    c.direct = 0;
    int cat = Ca / 16, cbt = Cb / 16;
    if (wgrad_direct_cfg(kind, bD, bH, bW, c)) { c.gy = cat * cbt; return c; }
    if (kind == 0) {
        c.pi = cat % 2 == 0 ? 2 : 1; c.pj = cbt % 2 == 0 ? 2 : 1;
        // One (ca, cb) pair per block with the 4 waves splitting K is what lets the next tile's loads ride in registers
        // (PREF); measured on the whole step: 5.15 ms with 2x2 pairs, 4.95 ms with single pairs at W >= 12.
        if (bW >= 12) { c.pi = 1; c.pj = 1; }
        if (bW >= 12) { c.bz = 2; c.by = 8; c.bx = 16; }
        else if (bW > 4) { c.bz = 4; c.by = 8; c.bx = 8; }
        else { c.bz = 4; c.by = 8; c.bx = 4; }
    } else {
        c.pi = 1; c.pj = cbt % 4 == 0 ? 4 : (cbt % 2 == 0 ? 2 : 1);
        if (bW >= 12) { c.bz = 2; c.by = 4; c.bx = 16; }
        else if (bW > 4) { c.bz = 2; c.by = 8; c.bx = 8; }
        else { c.bz = 4; c.by = 8; c.bx = 4; }
    }
    c.gy = (cat / c.pi) * (cbt / c.pj);
    int tiles = ((bW + c.bx - 1) / c.bx) * ((bH + c.by - 1) / c.by) * ((bD + c.bz - 1) / c.bz);
    int want = 512 / c.gy;
    if (want < 1) want = 1;
    c.nsplit = tiles < want ? tiles : want;
    return c;
}
// rows of the sliding-window stride-2 / conv_trans kernel (kernels_mfma_s2_wgrad.hip), 0 = not served
static int s2w_rows_conv(const ConvGeom& g) { return (g.ks == 3 && g.stride == 2) ? s2_wgrad_splits(g.Cin, g.Cout, g.Do, g.Ho, g.Wo) : 0; }
static int s2w_rows_convt(const ConvGeom& g) { return s2_wgrad_splits(g.Cout, g.Cin, g.D, g.H, g.W); }
size_t mfma_wgrad_scratch_bytes(const ConvGeom& g, int polite) {
    if (size_t z = mfma_wgrad_z_scratch_bytes(g, polite)) return z;
    if (int r = s2w_rows_conv(g)) return ((size_t)r * 27 * g.Cin * g.Cout + (size_t)r * g.Cout) * 4 + 256;
    WgradCfg c = wgrad_cfg(g.stride == 1 ? 0 : 1, g.Cin, g.Cout, g.Do, g.Ho, g.Wo);
    return ((size_t)c.nsplit * 27 * g.Cin * g.Cout + (size_t)c.nsplit * g.Cout) * 4 + 256;
}
size_t mfma_convt_wgrad_scratch_bytes(const ConvGeom& g) {
    if (int r = s2w_rows_convt(g)) return ((size_t)r * 8 * g.Cin * g.Cout + (size_t)r * g.Cout) * 4 + 256;
    WgradCfg c = wgrad_cfg(2, g.Cout, g.Cin, g.D, g.H, g.W);
    return ((size_t)c.nsplit * 8 * g.Cin * g.Cout + (size_t)c.nsplit * g.Cout) * 4 + 256;
}

template <int S, int KD, int PAD, int BZ, int BY, int BX, int PI, int PJ>
static void launch_wgrad_cfg(const MfmaWgradArgs& a0, const WgradCfg& c, hipStream_t s) {
    MfmaWgradArgs a = a0;
    a.tiles_x = (a.g.Wo + BX - 1) / BX; a.tiles_y = (a.g.Ho + BY - 1) / BY; a.tiles_z = (a.g.Do + BZ - 1) / BZ;
    constexpr int HZ = (BZ - 1) * S + KD, HY = (BY - 1) * S + KD, HX = (BX - 1) * S + KD, T = KD * KD * KD;
    constexpr size_t tile_lds = (size_t)PI * HZ * HY * HX * 32 + (size_t)PJ * BZ * BY * BX * 32;
    constexpr size_t red_lds = PI * PJ < 4 ? (size_t)(PI * PJ) * T * 64 * 16 : 0;   // only K-split waves reduce through LDS
    constexpr size_t stg_lds = (size_t)16 * (16 * T + 4) * 4;   // slab tile staging
    constexpr size_t lds0 = tile_lds > red_lds ? (tile_lds > 8192 ? tile_lds : 8192) : red_lds;
    constexpr size_t lds = lds0 > stg_lds ? lds0 : stg_lds;
    static_assert(lds <= 80 * 1024, "two blocks per CU");
    static std::atomic<uint64_t> attr_done{0};
    const int lds_launch = polite_lds((int)lds, c.polite);
    // the attribute is set once per kernel: to the polite size, whichever kind of launch comes first
    set_max_lds_once(attr_done, (const void*)k_mfma_wgrad<S, KD, PAD, BZ, BY, BX, PI, PJ>, polite_lds((int)lds, 1));
    dim3 grid((unsigned)c.nsplit, (unsigned)c.gy);
    a.direct = c.direct;
    k_mfma_wgrad<S, KD, PAD, BZ, BY, BX, PI, PJ><<<grid, 256, lds_launch, s>>>(a);
}
template <int S, int KD, int PAD, int BZ, int BY, int BX>
static void launch_wgrad_p(const MfmaWgradArgs& a, const WgradCfg& c, hipStream_t s) {
    if constexpr (S == 1) {
        if (c.pi == 2 && c.pj == 2) launch_wgrad_cfg<S, KD, PAD, BZ, BY, BX, 2, 2>(a, c, s);
        else if (c.pi == 2) launch_wgrad_cfg<S, KD, PAD, BZ, BY, BX, 2, 1>(a, c, s);
        else if (c.pj == 2) launch_wgrad_cfg<S, KD, PAD, BZ, BY, BX, 1, 2>(a, c, s);
        else launch_wgrad_cfg<S, KD, PAD, BZ, BY, BX, 1, 1>(a, c, s);
    } else {
        if (c.pj == 4) launch_wgrad_cfg<S, KD, PAD, BZ, BY, BX, 1, 4>(a, c, s);
        else if (c.pj == 2) launch_wgrad_cfg<S, KD, PAD, BZ, BY, BX, 1, 2>(a, c, s);
        else launch_wgrad_cfg<S, KD, PAD, BZ, BY, BX, 1, 1>(a, c, s);
    }
}

// rows of the slab a launch leaves at `scratch` ([rows][27*Cin*Cout], then [rows][Cout] bias partials): plan-time constant
int mfma_conv_wgrad_splits(const ConvGeom& g, int polite) {
    if (int z = mfma_wgrad_z_splits(g, polite)) return z;
    if (int r = s2w_rows_conv(g)) return r;
    return wgrad_cfg(g.stride == 1 ? 0 : 1, g.Cin, g.Cout, g.Do, g.Ho, g.Wo).nsplit;
}
int mfma_convt_wgrad_splits(const ConvGeom& g) {
    if (int r = s2w_rows_convt(g)) return r;
    return wgrad_cfg(2, g.Cout, g.Cin, g.D, g.H, g.W).nsplit;
}
// the launch adds into the gradient itself (no slab, nothing for a reduce pass to do)
bool mfma_conv_wgrad_direct(const ConvGeom& g) {
    if (mfma_wgrad_z_splits(g) || s2w_rows_conv(g)) return false;
    return wgrad_cfg(g.stride == 1 ? 0 : 1, g.Cin, g.Cout, g.Do, g.Ho, g.Wo).direct != 0;
}
bool mfma_convt_wgrad_direct(const ConvGeom& g) { return !s2w_rows_convt(g) && wgrad_cfg(2, g.Cout, g.Cin, g.D, g.H, g.W).direct != 0; }
// defer_reduce: leave the slab for the caller's batched reduce (launch_wgrad_reduce_batched) instead of summing it here
void launch_mfma_conv_wgrad(const ConvGeom& g, const SrcDesc* src, int nsrc, const void* dy, float* dw, float* db, void* scratch,
                            hipStream_t s, bool defer_reduce, int polite) {
    if (mfma_wgrad_z_supported(1, g, src, nsrc)) {   // sliding-window kernel (kernels_mfma_wgrad_z.hip): stride 1, W >= 24
        const int ns = launch_mfma_wgrad_z(g, src, nsrc, dy, db != nullptr, scratch, s, polite);
        const float* slab = (const float*)scratch;
        if (!defer_reduce)
            wgrad_reduce(slab, db ? slab + (size_t)ns * 27 * g.Cin * g.Cout : nullptr, ns, (int64_t)27 * g.Cin * g.Cout, g.Cout, dw, db, s);
        return;
    }
    if (s2w_rows_conv(g)) {     // sliding-window stride-2 kernel (kernels_mfma_s2_wgrad.hip): coarse grid >= 24 wide
        const int ns = launch_s2_wgrad(3, src[0].ptr, nsrc > 1 ? src[1].ptr : nullptr, src[0].C, g.Cin, g.D, g.H, g.W, dy, g.Cout, g.Do, g.Ho, g.Wo,
                                       db != nullptr, 0, scratch, s, polite);
        const float* slab = (const float*)scratch;
        if (!defer_reduce)
            wgrad_reduce(slab, db ? slab + (size_t)ns * 27 * g.Cin * g.Cout : nullptr, ns, (int64_t)27 * g.Cin * g.Cout, g.Cout, dw, db, s);
        return;
    }
    WgradCfg c = wgrad_cfg(g.stride == 1 ? 0 : 1, g.Cin, g.Cout, g.Do, g.Ho, g.Wo);
    c.polite = polite;
    MfmaWgradArgs a;
    a.g = g; a.nasrc = nsrc; a.asrc[0] = src[0]; if (nsrc > 1) a.asrc[1] = src[1];
    a.bsrc = SrcDesc(); a.bsrc.ptr = dy; a.bsrc.C = g.Cout;
    a.bias_from_a = 0;
    a.slab = (float*)scratch;
    a.bias_slab = db ? a.slab + (size_t)c.nsplit * 27 * g.Cin * g.Cout : nullptr;
    a.tiles_x = a.tiles_y = a.tiles_z = 0;
    if (c.direct) {   // the block adds into dw / db themselves
        a.slab = dw; a.bias_slab = db;
        if (g.stride == 1 && c.bx == 4) launch_wgrad_cfg<1, 3, 1, 4, 4, 4, 1, 1>(a, c, s);
        else if (g.stride == 1) launch_wgrad_cfg<1, 3, 1, 8, 8, 8, 1, 1>(a, c, s);
        else if (c.bx == 4) launch_wgrad_cfg<2, 3, 1, 4, 4, 4, 1, 1>(a, c, s);
        else launch_wgrad_cfg<2, 3, 1, 2, 8, 8, 1, 1>(a, c, s);
        return;
    }
    if (g.stride == 1) {
        if (g.Wo >= 12) launch_wgrad_p<1, 3, 1, 2, 8, 16>(a, c, s);
        else if (g.Wo > 4) launch_wgrad_p<1, 3, 1, 4, 8, 8>(a, c, s);
        else launch_wgrad_p<1, 3, 1, 4, 8, 4>(a, c, s);
    } else {
        if (g.Wo >= 12) launch_wgrad_p<2, 3, 1, 2, 4, 16>(a, c, s);
        else if (g.Wo > 4) launch_wgrad_p<2, 3, 1, 2, 8, 8>(a, c, s);
        else launch_wgrad_p<2, 3, 1, 4, 8, 4>(a, c, s);
    }
    if (!defer_reduce) wgrad_reduce(a.slab, a.bias_slab, c.nsplit, (int64_t)27 * g.Cin * g.Cout, g.Cout, dw, db, s);
}

// conv_trans wgrad (g = forward geometry of the conv_trans: D,H,W coarse input, Do,Ho,Wo fine output).
// halo side A = dy (fine, Cout channels), tile side B = transformed input (coarse, Cin channels): D_t[co][ci].
void launch_mfma_convt_wgrad(const ConvGeom& g, const SrcDesc* src, const void* dy, float* dw, void* scratch, hipStream_t s, bool defer_reduce,
                             float* db, int polite) {
    if (s2w_rows_convt(g)) {    // fine = dy (Ca = Cout), coarse = the input (Cb = Cin): slab [Cin][Cout][8] = the layout of dw; bias = sums of dy
        const int ns = launch_s2_wgrad(2, dy, nullptr, g.Cout, g.Cout, g.Do, g.Ho, g.Wo, src[0].ptr, g.Cin, g.D, g.H, g.W, db != nullptr, 1, scratch, s, polite);
        const float* slab = (const float*)scratch;
        if (!defer_reduce)
            wgrad_reduce(slab, db ? slab + (size_t)ns * 8 * g.Cin * g.Cout : nullptr, ns, (int64_t)8 * g.Cin * g.Cout, g.Cout, dw, db, s);
        return;
    }
    WgradCfg c = wgrad_cfg(2, g.Cout, g.Cin, g.D, g.H, g.W);
    c.polite = polite;
    MfmaWgradArgs a;
    a.bias_from_a = 1;
    a.g.Cin = g.Cout; a.g.Cout = g.Cin; a.g.D = g.Do; a.g.H = g.Ho; a.g.W = g.Wo; a.g.Do = g.D; a.g.Ho = g.H; a.g.Wo = g.W;
    a.g.ks = 2; a.g.stride = 2;
    a.nasrc = 1; a.asrc[0] = SrcDesc(); a.asrc[0].ptr = dy; a.asrc[0].C = g.Cout;
    a.bsrc = src[0];
    a.slab = (float*)scratch;
    // the bias gradient rides along (sums of dy as it is staged): [nsplit][Cout] behind the slab
    a.bias_slab = db ? a.slab + (size_t)c.nsplit * 8 * g.Cin * g.Cout : nullptr;
    a.tiles_x = a.tiles_y = a.tiles_z = 0;
    if (c.direct) {
        a.slab = dw; a.bias_slab = db;
        if (c.bx == 4) launch_wgrad_cfg<2, 2, 0, 4, 4, 4, 1, 1>(a, c, s);
        else launch_wgrad_cfg<2, 2, 0, 2, 8, 8, 1, 1>(a, c, s);
        return;
    }
    if (g.W >= 12) launch_wgrad_p<2, 2, 0, 2, 4, 16>(a, c, s);
    else if (g.W > 4) launch_wgrad_p<2, 2, 0, 2, 8, 8>(a, c, s);
    else launch_wgrad_p<2, 2, 0, 4, 8, 4>(a, c, s);
    // slab[cb = ci][ca = co][t] = the layout of dw ([Cin][Cout][2][2][2])
    if (!defer_reduce) wgrad_reduce(a.slab, a.bias_slab, c.nsplit, (int64_t)8 * g.Cin * g.Cout, g.Cout, dw, db, s);
}

}  // namespace unet
