// fp32 3x3x3 convolution (stride 1 and 2) on the fp32 matrix cores (v_mfma_f32_16x16x4_f32), for the fp32 engine
// (UNET_DTYPE_F32: the parity configuration and BASELINE.json configs[1], the full forward on a 128^3 volume in fp32).
// The products and the accumulation are IEEE fp32, exactly as in the VALU kernel of kernels_direct.hip it replaces when
// impl == AUTO -- only the summation order differs -- so the 1e-4 logit tolerance of the fp32 path holds unchanged.
//
//   D[voxel x][cout] += X[voxel x + kx][cin] * W[tap][cin][cout]        M = 16 voxels of one x-row, N = 16 cout, K = 4 cin
//
// Block = 256 threads = 4 waves, output tile 4 (z) x BY (y) x 16 (x) voxels x 16*NT output channels; wave w owns the BY rows of
// z-slice w.  Per CK-channel chunk of the input: the halo tile (stride 1, BY 8: 6 x 10 x 18 voxels; stride 2, BY 2: 9 x 5 x 33) is
// staged channel-major ([cin][voxel], plane stride = 16 mod 64 floats, so the four k-groups of a wave hit four disjoint bank
// ranges) with the consumer-side transform act(x*scale+shift) applied on the way in (zero padding is applied AFTER it, as the
// reference pads the activated tensor); the chunk's filter slice is staged as [nt][tap][cin][16].  For one (kz, kx, k-group) a
// wave reads the halo rows it needs ONCE and feeds them to 3 (ky) x BY (rows) x NT MFMAs (stride 1: 10 + 3*NT LDS reads per
// 24*NT MFMAs).  Measured at 128^3: 32->16 (58 GFLOP) in 0.60 ms = 97 TFLOP/s of the 157 the fp32 matrix pipe has.
#include <cstdlib>
#include "mfma_util.h"
#include "kernels.h"

namespace unet {

namespace {

constexpr int F_BZ = 4, F_BX = 16;
template <int S, int BY, int CK> struct F32Tile {
    static constexpr int HZ = (F_BZ - 1) * S + 3, HY = (BY - 1) * S + 3, HX = (F_BX - 1) * S + 3;
    static constexpr int HV = HZ * HY * HX;                       // halo voxels
    static constexpr int PS = (HV + 47) / 64 * 64 + 16;           // plane stride in floats: >= HV and == 16 (mod 64)
    static_assert(PS >= HV && PS % 64 == 16, "bank layout");
    static constexpr size_t lds_bytes(int NT) { return (size_t)(CK * PS + NT * 27 * CK * 16) * sizeof(float); }
};

struct ConvF32Args {
    ConvGeom g;
    SrcDesc s0, s1;          // up to two concatenated sources (skip first, unet.cpp:181)
    const float* w;          // [27][Cin][CoutP]  (launch_pack_conv_w)
    const float* bias;
    float* out;              // channels-last fp32 [voxel][Cout]   (forward)
    // dgrad: the kernel runs on dy with the tap order reversed and the [tap][cout][cinP] filter copy; its "output channels" are the
    // layer's input channels, which go to up to two gradient tensors (the two concatenated sources), written or accumulated
    int flip;                // 1: filter tap = 26 - tap
    float* dst[2];
    int dstC[2], dst_acc[2];
    int CoutP;
    int tz, ty, tx;          // tile grid
    // forward only: per-block {sum, sum of squares} of the block's OUTPUT values per channel, [blockIdx.x][Cout][2] in fp64 -- what
    // k_stats_partial<float, 0> leaves for an fp32 tensor (ATen's CPU norm kernels accumulate in double, acc_type<float>), so the
    // norm layer that follows needs no pass of its own over the tensor
    double* stats;
};

template <int S, int BY, int CK, int NT> __global__ void __launch_bounds__(256, 2) k_conv_f32_mfma(ConvF32Args a) {
    typedef F32Tile<S, BY, CK> TL;
    constexpr int F_BY = BY, F_CK = CK, F_HY = TL::HY, F_HX = TL::HX, F_HV = TL::HV, F_PS = TL::PS;
    extern __shared__ float lds[];
    float* xs = lds;                          // [F_CK][F_PS]
    float* wsm = lds + F_CK * F_PS;           // [NT][27][F_CK][16]
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, lx = lane & 15, lq = lane >> 4;
    int t = xcd_remap(blockIdx.x, gridDim.x);
    const int bx = t % a.tx; t /= a.tx;
    const int by = t % a.ty, bz = t / a.ty;
    const int z0 = bz * F_BZ, y0 = by * F_BY, x0 = bx * F_BX;
    const int co0 = blockIdx.y * (16 * NT);

    f32x4 acc[F_BY][NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const float b = a.bias ? a.bias[co0 + n * 16 + lx] : 0.f;
#pragma unroll
        for (int r = 0; r < F_BY; ++r) acc[r][n] = f32x4{b, b, b, b};
    }

    // The next chunk's global loads ride in registers through this chunk's MFMAs (round 3): the loop used to be load -> LDS ->
    // barrier -> 432 MFMAs per chunk with every load latency exposed (~2 of ~9 us per chunk and block at 128^3).
    constexpr int ITX = (F_HV * (F_CK / 4) + 255) / 256;        // halo float4 per thread and chunk
    constexpr int ITW = (27 * F_CK * 4 * NT + 255) / 256;       // filter float4 per thread and chunk
    float4 XR[ITX], WR[ITW];
    unsigned xin = 0;                                           // bit i: unit i lies inside the volume (zero padding applies AFTER the transform)
    auto fetch = [&](int c0) {
        const bool second = c0 >= a.s0.C;
        const float* sp = (const float*)(second ? a.s1.ptr : a.s0.ptr);
        const int sC = second ? a.s1.C : a.s0.C, cb = second ? c0 - a.s0.C : c0;
        xin = 0;
#pragma unroll
        for (int i = 0; i < ITX; ++i) {
            const int it = tid + i * 256;
            const int hv = it / (F_CK / 4), q = it % (F_CK / 4);
            const int hz = hv / (F_HY * F_HX), rem = hv - hz * (F_HY * F_HX), hy = rem / F_HX, hx = rem - hy * F_HX;
            const int iz = z0 * S + hz - 1, iy = y0 * S + hy - 1, ix = x0 * S + hx - 1;
            XR[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (it < F_HV * (F_CK / 4) && iz >= 0 && iz < g.D && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) {
                XR[i] = *(const float4*)(sp + (((int64_t)iz * g.H + iy) * g.W + ix) * sC + cb + q * 4);
                xin |= 1u << i;
            }
        }
#pragma unroll
        for (int i = 0; i < ITW; ++i) {
            const int it = tid + i * 256;
            const int c4 = it % (4 * NT), rk = it / (4 * NT), k = rk % F_CK, tap = rk / F_CK;
            WR[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (it < 27 * F_CK * 4 * NT)
                WR[i] = *(const float4*)(a.w + ((int64_t)(a.flip ? 26 - tap : tap) * g.Cin + c0 + k) * a.CoutP + co0 + c4 * 4);
        }
    };
    auto commit = [&](int c0) {
        const bool second = c0 >= a.s0.C;
        const int cb = second ? c0 - a.s0.C : c0;
        const float* sc = second ? a.s1.scale : a.s0.scale;
        const float* sh = second ? a.s1.shift : a.s0.shift;
        const int act = second ? a.s1.act : a.s0.act;
#pragma unroll
        for (int i = 0; i < ITX; ++i) {
            const int it = tid + i * 256;
            if (it >= F_HV * (F_CK / 4)) continue;
            const int hv = it / (F_CK / 4), q = it % (F_CK / 4);
            float4 v = XR[i];
            if ((xin >> i) & 1u) {
                const int c = cb + q * 4;
                if (sc) {
                    const float4 s4 = *(const float4*)(sc + c), h4 = *(const float4*)(sh + c);
                    v.x = v.x * s4.x + h4.x; v.y = v.y * s4.y + h4.y; v.z = v.z * s4.z + h4.z; v.w = v.w * s4.w + h4.w;
                }
                v.x = act_f(v.x, act); v.y = act_f(v.y, act); v.z = act_f(v.z, act); v.w = act_f(v.w, act);
            }
            float* d = xs + (q * 4) * F_PS + hv;
            d[0] = v.x; d[F_PS] = v.y; d[2 * F_PS] = v.z; d[3 * F_PS] = v.w;
        }
#pragma unroll
        for (int i = 0; i < ITW; ++i) {
            const int it = tid + i * 256;
            if (it >= 27 * F_CK * 4 * NT) continue;
            const int c4 = it % (4 * NT), rk = it / (4 * NT), k = rk % F_CK, tap = rk / F_CK;
            const int n = c4 >> 2, cc = (c4 & 3) * 4;
            *(float4*)(wsm + ((n * 27 + tap) * F_CK + k) * 16 + cc) = WR[i];
        }
    };
    fetch(0);
    for (int c0 = 0; c0 < g.Cin; c0 += F_CK) {
        __syncthreads();   // the previous chunk's MFMAs have read xs / wsm
        commit(c0);
        __syncthreads();
        if (c0 + F_CK < g.Cin) fetch(c0 + F_CK);     // in flight during the MFMAs below
#pragma unroll
        for (int kz = 0; kz < 3; ++kz)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int kg = 0; kg < F_CK / 4; ++kg) {
                    float A[F_HY];
                    const float* xp = xs + (kg * 4 + lq) * F_PS + ((wv * S + kz) * F_HY) * F_HX + lx * S + kx;
#pragma unroll
                    for (int i = 0; i < F_HY; ++i) A[i] = xp[i * F_HX];
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            const float b = wsm[((n * 27 + (kz * 3 + ky) * 3 + kx) * F_CK + kg * 4 + lq) * 16 + lx];
#pragma unroll
                            for (int r = 0; r < F_BY; ++r)
                                acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * S + ky], b, acc[r][n], 0, 0, 0);
                        }
                }
    }
    // C layout of the 16x16 tile: lane (lq, lx) holds rows (= voxels x) 4*lq + j, column (= cout) lx
    const int z = z0 + wv;
    // destination of this block's channel tile (a tile never straddles two gradient tensors: launch_conv_f32_mfma_dgrad picks NT)
    float* ob = a.out;
    int oC = g.Cout, ocb = co0, oacc = 0;
    if (a.flip) {
        const int second = co0 >= a.dstC[0] ? 1 : 0;
        ob = second ? a.dst[1] : a.dst[0];
        oC = second ? a.dstC[1] : a.dstC[0];
        ocb = co0 - (second ? a.dstC[0] : 0);
        oacc = second ? a.dst_acc[1] : a.dst_acc[0];
    }
    if (a.stats) {   // wave-uniform
        __syncthreads();                       // every wave is done with the tiles in LDS
        double* red = (double*)lds;            // [4 waves][NT][16 co][2]
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int r = 0; r < F_BY; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (z < g.Do && y0 + r < g.Ho && x0 + 4 * lq + j < g.Wo) {
                        const double v = (double)acc[r][n][j];
                        s1 += v; s2 += v * v;
                    }
            s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);      // the four voxel groups (lq) of a channel (lx)
            s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
            if (lq == 0) { red[((wv * NT + n) * 16 + lx) * 2] = s1; red[((wv * NT + n) * 16 + lx) * 2 + 1] = s2; }
        }
        __syncthreads();
        if (tid < NT * 16) {
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { s1 += red[(w * NT * 16 + tid) * 2]; s2 += red[(w * NT * 16 + tid) * 2 + 1]; }
            double* o = a.stats + ((int64_t)blockIdx.x * g.Cout + co0 + tid) * 2;
            o[0] = s1; o[1] = s2;
        }
    }
    if (z < g.Do && ob) {
#pragma unroll
        for (int r = 0; r < F_BY; ++r) {
            const int y = y0 + r;
            if (y >= g.Ho) continue;
            float* o = ob + (((int64_t)z * g.Ho + y) * g.Wo) * oC + ocb + lx;
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int x = x0 + 4 * lq + j;
                    if (x < g.Wo) {
                        float* q = o + (int64_t)x * oC + n * 16;
                        *q = oacc ? *q + acc[r][n][j] : acc[r][n][j];
                    }
                }
        }
    }
}

}  // namespace

bool conv_f32_mfma_supported(int dtype, const ConvGeom& g, const SrcDesc* src, int nsrc) {
    if (dtype != 0 || g.ks != 3 || (g.stride != 1 && g.stride != 2) || g.Cout % 16 || g.Cin % 8) return false;
    return nsrc >= 1 && nsrc <= 2 && src[0].C % 8 == 0 && (nsrc == 1 || src[1].C % 8 == 0);
}

template <int S, int BY, int CK, int NT> static void launch_f32_variant(ConvF32Args& a, hipStream_t s) {
    typedef F32Tile<S, BY, CK> TL;
    const size_t lds = TL::lds_bytes(NT);
    static std::atomic<uint64_t> once{0};   // > 64 KB of dynamic LDS needs the opt-in, per device
    set_max_lds_once(once, (const void*)k_conv_f32_mfma<S, BY, CK, NT>, (int)lds);
    a.tz = (a.g.Do + F_BZ - 1) / F_BZ; a.ty = (a.g.Ho + BY - 1) / BY; a.tx = (a.g.Wo + F_BX - 1) / F_BX;
    k_conv_f32_mfma<S, BY, CK, NT><<<dim3((unsigned)(a.tz * a.ty * a.tx), a.g.Cout / (16 * NT)), 256, lds, s>>>(a);
}

// which tile the forward uses: 0 = 4x8x16 / 8-channel chunks, 1 = the same with 32 output channels per block, 2 = 4x4x16 / 16-channel
// chunks (levels with few tiles), 3 / 4 = stride 2 (4x2x16, 16 / 32 output channels per block)
static int f32_fwd_variant(const ConvGeom& g, const SrcDesc* src) {
    // 32 output channels per block halve the input staging; small volumes take 16 so that twice as many blocks share the walk over Cin
    const int by = g.stride == 1 ? 8 : 2;
    const int64_t tiles = (int64_t)((g.Do + 3) / 4) * ((g.Ho + by - 1) / by) * ((g.Wo + 15) / 16);
    const bool wide = g.Cout % 32 == 0 && tiles * (g.Cout / 32) >= 512;
    // levels with few tiles (32^3 and below in the default architecture): a 4x4x16 tile with 16-channel chunks -- twice the blocks,
    // half the chunk passes (forward at 128^3: 8.4 -> 7.4 ms); the 128^3 layers are faster on the 4x8x16 / 8-channel form
    // (0.62 vs 0.68 ms for 32->16; profiles/r09_forward_kernel_stats.csv)
    const bool few = tiles * (g.Cout / 16) < 2048;
    if (g.stride == 1 && !wide && g.Cin % 16 == 0 && src[0].C % 16 == 0 && few) return 2;
    if (g.stride == 1) return wide ? 1 : 0;
    // stride 2: 4x2x16 outputs from a 9x5x33 halo in 8-channel chunks (0.37 ms per forward faster than 4x4x16 with 4-channel chunks,
    // whose halo forced twice the chunk passes)
    return wide ? 4 : 3;
}
// rows of the statistics partials the forward leaves (one per tile): plan-time constant
int conv_f32_mfma_stat_rows(const ConvGeom& g, const SrcDesc* src) {
    const int v = f32_fwd_variant(g, src);
    const int by = v == 2 ? 4 : (v >= 3 ? 2 : 8);
    return ((g.Do + 3) / 4) * ((g.Ho + by - 1) / by) * ((g.Wo + 15) / 16);
}
int launch_conv_f32_mfma(const ConvGeom& g, const SrcDesc* src, int nsrc, const float* w_fwd, const float* bias, float* out,
                         hipStream_t s, double* stats_partial) {
    ConvF32Args a;
    a.g = g; a.s0 = src[0]; a.s1 = nsrc > 1 ? src[1] : SrcDesc();
    if (nsrc == 1) a.s0.C = g.Cin;
    a.w = w_fwd; a.bias = bias; a.out = out; a.CoutP = round_up(g.Cout, 8);
    a.flip = 0; a.dst[0] = a.dst[1] = nullptr; a.dstC[0] = a.dstC[1] = 0; a.dst_acc[0] = a.dst_acc[1] = 0;
    a.stats = stats_partial;
    switch (f32_fwd_variant(g, src)) {
        case 2: launch_f32_variant<1, 4, 16, 1>(a, s); break;
        case 1: launch_f32_variant<1, 8, 8, 2>(a, s); break;
        case 0: launch_f32_variant<1, 8, 8, 1>(a, s); break;
        case 4: launch_f32_variant<2, 2, 8, 2>(a, s); break;
        default: launch_f32_variant<2, 2, 8, 1>(a, s); break;
    }
    return a.tz * a.ty * a.tx;
}

// ================================================================================================================
// The network's first conv in fp32 (Cin = 1, 3x3x3, stride 1, Cout = 16 * NT) on the fp32 matrix cores: K = the 27 taps padded to 28,
//     y[voxel][co] = bias[co] + sum_tap x[voxel + tap] * w[co][tap]          M = 16 voxels of one x-row, N = 16 co, K = 4 taps
// Tile 4 x 8 x 16 voxels, halo 6 x 10 x 18 floats in LDS; a lane gathers its tap of the k-step with one 4-byte LDS read, the 7 x NT
// filter operands sit in registers.  Output-bound (64 B per voxel written for 4 B read); replaces the VALU k_conv_first<float> (0.20 ms
// of the fp32 forward at 128^3) and leaves the norm statistics of its output (fp64 rows per block) like k_conv_f32_mfma.
// ================================================================================================================
namespace {
struct ConvFirstF32Args {
    ConvGeom g;
    const float* x;      // [D][H][W]
    const float* w;      // [Cout][27] (torch layout, Cin = 1)
    const float* bias;
    float* out;          // [D][H][W][Cout]
    double* stats;       // [gridDim.x][Cout][2] or nullptr
    int tiles_x, tiles_y, tiles_z;
};
template <int NT> __global__ void __launch_bounds__(256) k_conv_first_f32_mfma(ConvFirstF32Args a) {
    constexpr int BZ = 4, BY = 8, BX = 16, HZ = BZ + 2, HY = BY + 2, HX = BX + 2, NV = HZ * HY * HX, ITERS = (NV + 255) / 256;
    __shared__ float tile[NV];
    __shared__ double red[4 * NT * 16 * 2];
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lx = lane & 15, lq = lane >> 4;
    const int ntiles = a.tiles_x * a.tiles_y * a.tiles_z;
    // B operands: lane (k = lq, col = lx) of k-step s <- w[co = n*16 + lx][tap = 4 s + lq]; A addresses: tap -> offset inside the tile
    float wb[7][NT];
    int aoff[7];
#pragma unroll
    for (int st = 0; st < 7; ++st) {
        const int tap = 4 * st + lq;
#pragma unroll
        for (int n = 0; n < NT; ++n) wb[st][n] = tap < 27 ? a.w[(n * 16 + lx) * 27 + tap] : 0.f;
        const int tp = tap < 27 ? tap : 26;     // the padding tap reads a finite value (times a zero filter operand)
        aoff[st] = (((tp / 9) + wave) * HY + (tp / 3) % 3) * HX + tp % 3 + lx;
    }
    float bcol[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) bcol[n] = a.bias ? a.bias[n * 16 + lx] : 0.f;
    double s1[NT], s2[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) { s1[n] = 0.0; s2[n] = 0.0; }
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int x0 = (t % a.tiles_x) * BX, y0 = ((t / a.tiles_x) % a.tiles_y) * BY, z0 = (t / (a.tiles_x * a.tiles_y)) * BZ;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < ITERS; ++k) {
            const int u = tid + k * 256;
            if (u < NV) {
                const int hz = u / (HY * HX), hr = u % (HY * HX), hy = hr / HX, hx = hr % HX;
                const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
                tile[u] = ((unsigned)gz < (unsigned)g.D && (unsigned)gy < (unsigned)g.H && (unsigned)gx < (unsigned)g.W)
                              ? a.x[((size_t)gz * g.H + gy) * g.W + gx] : 0.f;
            }
        }
        __syncthreads();
        const int gz = z0 + wave;
#pragma unroll
        for (int i = 0; i < BY; ++i) {            // m-tile = row i of z-plane `wave`: 16 voxels along x
            f32x4 acc[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[n] = f32x4{bcol[n], bcol[n], bcol[n], bcol[n]};
#pragma unroll
            for (int st = 0; st < 7; ++st) {
                const float av = tile[aoff[st] + i * HX];
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wb[st][n], acc[n], 0, 0, 0);
            }
            const int gy = y0 + i;
            if (gz < g.D && gy < g.H) {
                float* o = a.out + (((size_t)gz * g.H + gy) * g.W) * g.Cout + lx;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int gx = x0 + 4 * lq + j;
                    if (gx < g.W) {
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            const float v = acc[n][j];
                            o[(size_t)gx * g.Cout + n * 16] = v;
                            s1[n] += (double)v; s2[n] += (double)v * (double)v;
                        }
                    }
                }
            }
        }
    }
    if (a.stats) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            double u = s1[n], v = s2[n];
            u += __shfl_xor(u, 16); v += __shfl_xor(v, 16);
            u += __shfl_xor(u, 32); v += __shfl_xor(v, 32);
            if (lq == 0) { red[((wave * NT + n) * 16 + lx) * 2] = u; red[((wave * NT + n) * 16 + lx) * 2 + 1] = v; }
        }
        __syncthreads();
        if (tid < NT * 16) {
            double u = 0.0, v = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { u += red[(w * NT * 16 + tid) * 2]; v += red[(w * NT * 16 + tid) * 2 + 1]; }
            a.stats[((size_t)blockIdx.x * g.Cout + tid) * 2] = u;
            a.stats[((size_t)blockIdx.x * g.Cout + tid) * 2 + 1] = v;
        }
    }
}
}  // namespace

bool conv_first_f32_mfma_supported(int dtype, const ConvGeom& g, const SrcDesc* src, int nsrc) {
    return dtype == 0 && nsrc == 1 && g.Cin == 1 && g.ks == 3 && g.stride == 1 && (g.Cout == 16 || g.Cout == 32) && !src[0].scale &&
           src[0].act == 0 && src[0].C == 1;
}
int conv_first_f32_mfma_blocks(const ConvGeom& g) {
    const int tiles = ((g.W + 15) / 16) * ((g.H + 7) / 8) * ((g.D + 3) / 4);
    return tiles < 1024 ? tiles : 1024;
}
// returns the number of statistics rows written (one per block) when stats_partial is given
int launch_conv_first_f32_mfma(const ConvGeom& g, const SrcDesc* src, const float* w, const float* bias, float* out, double* stats_partial,
                               hipStream_t s) {
    ConvFirstF32Args a;
    a.g = g; a.x = (const float*)src[0].ptr; a.w = w; a.bias = bias; a.out = out; a.stats = stats_partial;
    a.tiles_x = (g.W + 15) / 16; a.tiles_y = (g.H + 7) / 8; a.tiles_z = (g.D + 3) / 4;
    const int nb = conv_first_f32_mfma_blocks(g);
    if (g.Cout == 16) k_conv_first_f32_mfma<1><<<nb, 256, 0, s>>>(a);
    else k_conv_first_f32_mfma<2><<<nb, 256, 0, s>>>(a);
    return nb;
}

// ================================================================================================================
// fp32 ConvTranspose3d(k2, s2) forward (unet.cpp:46-57) on the fp32 matrix cores: 8 independent 1x1 GEMMs, one per output parity,
//     y[2v + t][co] = bias[co] + sum_ci x[v][ci] * W[ci][co][t]          M = 16 voxels of one x-row, N = 16 co, K = 4 ci
// Block = 4 waves = 4 input rows (y) x 16 voxels (x) x 16 output channels x all 8 taps; the input rows are staged channel-major
// ([ci][row][x], with act(x*scale+shift) applied on the way in), the filter slice [tap][ci][16 co] beside them, in 32-channel chunks.
// Write-bound (8 x Cout floats out per Cin floats in): the VALU kernel it replaces (k_convt_fwd_direct) held 1.1 ms of the 6.8-ms
// fp32 forward at 128^3 for 1.7 % of its FLOPs.
// ================================================================================================================
namespace {
constexpr int CT_CK = 32, CT_PS = 64 + 16;   // channel chunk; plane stride in floats (64 voxels, == 16 mod 64: the four k-groups hit disjoint banks)
struct ConvtF32Args {
    ConvGeom g;
    SrcDesc src;
    const float* w;          // [8][Cin][CoutP]  (launch_pack_convt_w)
    const float* bias;
    float* out;              // channels-last fp32 [2D][2H][2W][Cout]
    int CoutP, tx;           // x tiles per row
};
__global__ void __launch_bounds__(256) k_convt_f32_mfma(ConvtF32Args a) {
    __shared__ float xs[CT_CK * CT_PS];          // [ci][row * 16 + x]
    __shared__ float wsm[8 * CT_CK * 16];        // [tap][ci][16 co]
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, lx = lane & 15, lq = lane >> 4;
    const int bx = (int)blockIdx.x % a.tx;
    const int64_t rowg = (int64_t)blockIdx.x / a.tx;          // group of 4 consecutive (z, y) rows of the input volume
    const int x0 = bx * 16, co0 = blockIdx.y * 16;
    const int64_t nrows = (int64_t)g.D * g.H;
    f32x4 acc[8];
    {
        const float b = a.bias ? a.bias[co0 + lx] : 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = f32x4{b, b, b, b};
    }
    const float* sp = (const float*)a.src.ptr;
    for (int c0 = 0; c0 < g.Cin; c0 += CT_CK) {
        __syncthreads();
        // 64 voxels x 32 channels: a thread loads float4 = 4 channels of one voxel
        for (int it = tid; it < 64 * (CT_CK / 4); it += 256) {
            const int v = it / (CT_CK / 4), q = it % (CT_CK / 4);
            const int64_t row = rowg * 4 + (v >> 4);
            const int x = x0 + (v & 15);
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            const int c = c0 + q * 4;
            if (row < nrows && x < g.W && c < g.Cin) {
                val = *(const float4*)(sp + (row * g.W + x) * (int64_t)a.src.C + c);
                if (a.src.scale) {
                    const float4 s4 = *(const float4*)(a.src.scale + c), h4 = *(const float4*)(a.src.shift + c);
                    val.x = val.x * s4.x + h4.x; val.y = val.y * s4.y + h4.y; val.z = val.z * s4.z + h4.z; val.w = val.w * s4.w + h4.w;
                }
                val.x = act_f(val.x, a.src.act); val.y = act_f(val.y, a.src.act); val.z = act_f(val.z, a.src.act); val.w = act_f(val.w, a.src.act);
            }
            float* d = xs + (q * 4) * CT_PS + v;
            d[0] = val.x; d[CT_PS] = val.y; d[2 * CT_PS] = val.z; d[3 * CT_PS] = val.w;
        }
        for (int it = tid; it < 8 * CT_CK * 4; it += 256) {    // float4 = 4 co of one (tap, ci)
            const int c4 = it & 3, k = (it >> 2) % CT_CK, tap = it / (4 * CT_CK);
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c0 + k < g.Cin) val = *(const float4*)(a.w + ((int64_t)tap * g.Cin + c0 + k) * a.CoutP + co0 + c4 * 4);
            *(float4*)(wsm + (tap * CT_CK + k) * 16 + c4 * 4) = val;
        }
        __syncthreads();
#pragma unroll
        for (int kg = 0; kg < CT_CK / 4; ++kg) {
            const float A = xs[(kg * 4 + lq) * CT_PS + wv * 16 + lx];
#pragma unroll
            for (int t = 0; t < 8; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A, wsm[(t * CT_CK + kg * 4 + lq) * 16 + lx], acc[t], 0, 0, 0);
        }
    }
    // lane (lq, lx): rows = voxels x0 + 4*lq + j, column = co0 + lx
    const int64_t row = rowg * 4 + wv;
    if (row >= nrows) return;
    const int z = (int)(row / g.H), y = (int)(row % g.H);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int tz = t >> 2, ty = (t >> 1) & 1, tx = t & 1;
        float* o = a.out + ((((int64_t)(2 * z + tz) * g.Ho + (2 * y + ty)) * g.Wo) * g.Cout) + co0 + lx;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = x0 + 4 * lq + j;
            if (x < g.W) o[(int64_t)(2 * x + tx) * g.Cout] = acc[t][j];
        }
    }
}
}  // namespace

bool convt_f32_mfma_supported(int dtype, const ConvGeom& g, const SrcDesc* src, int nsrc) {
    return dtype == 0 && nsrc == 1 && g.Cin % 4 == 0 && g.Cout % 16 == 0 && src[0].C == g.Cin && g.Do == 2 * g.D && g.Ho == 2 * g.H && g.Wo == 2 * g.W;
}
void launch_convt_f32_mfma(const ConvGeom& g, const SrcDesc* src, const float* w_fwd, const float* bias, float* out, hipStream_t s) {
    ConvtF32Args a;
    a.g = g; a.src = src[0]; a.w = w_fwd; a.bias = bias; a.out = out; a.CoutP = round_up(g.Cout, 8);
    a.tx = (g.W + 15) / 16;
    const int64_t rowgroups = ((int64_t)g.D * g.H + 3) / 4;
    k_convt_f32_mfma<<<dim3((unsigned)(rowgroups * a.tx), (unsigned)(g.Cout / 16)), 256, 0, s>>>(a);
}

// ---- input gradient of a stride-1 3x3x3 conv: dx[u][ci] = sum_k sum_co dy[u + 1 - k][co] * w[co][ci][k] ----
bool conv_f32_mfma_dgrad_supported(int dtype, const ConvGeom& g, const DstGrad* dst, int ndst) {
    if (dtype != 0 || g.ks != 3 || g.stride != 1 || g.Cout % 8 || g.Cin % 16 || ndst < 1 || ndst > 2) return false;
    int c = 0;
    for (int k = 0; k < ndst; ++k) { if (dst[k].C % 16) return false; c += dst[k].C; }
    return c == g.Cin;
}

void launch_conv_f32_mfma_dgrad(const ConvGeom& g, const float* dy, const float* w_dgrad, const DstGrad* dst, int ndst, hipStream_t s) {
    ConvF32Args a;
    a.g = g;                      // the kernel's view: input = dy (Cout channels), output = dx (Cin channels), same volume
    a.g.Cin = g.Cout; a.g.Cout = g.Cin; a.g.D = g.Do; a.g.H = g.Ho; a.g.W = g.Wo;
    a.g.Do = g.D; a.g.Ho = g.H; a.g.Wo = g.W;
    a.s0 = SrcDesc(); a.s0.ptr = dy; a.s0.C = g.Cout; a.s1 = SrcDesc();
    a.w = w_dgrad; a.bias = nullptr; a.out = nullptr; a.CoutP = round_up(g.Cin, 8);
    a.flip = 1; a.stats = nullptr;
    for (int k = 0; k < 2; ++k) {
        a.dst[k] = k < ndst ? (float*)dst[k].ptr : nullptr;
        a.dstC[k] = k < ndst ? dst[k].C : 0;
        a.dst_acc[k] = k < ndst ? dst[k].accumulate : 0;
    }
    // 32-channel tiles only when no tile can straddle the two gradient tensors
    const bool nt2_ok = g.Cin % 32 == 0 && (ndst == 1 || dst[0].C % 32 == 0);
    const int64_t tiles = (int64_t)((g.D + 3) / 4) * ((g.H + 7) / 8) * ((g.W + 15) / 16);
    const bool wide = nt2_ok && tiles * (g.Cin / 32) >= 512;
    const bool few = tiles * (g.Cin / 16) < 2048;
    if (wide) launch_f32_variant<1, 8, 8, 2>(a, s);
    else if (few && g.Cout % 16 == 0) launch_f32_variant<1, 4, 16, 1>(a, s);
    else launch_f32_variant<1, 8, 8, 1>(a, s);
}

// ================================================================================================================
// fp32 weight gradient of a stride-1 3x3x3 conv on the fp32 matrix cores:
//   dW[tap][ci][co] = sum_v x[v + tap - 1][ci] * dy[v][co]        M = 16 ci, N = 16 co, K = 4 voxels of one x-row
// grid = (persistent blocks over 2x8x16-voxel tiles, Cin/16, Cout/16); a block stages the 4x10x18 halo of its 16 input channels
// ([voxel][16 ch]: the 4 voxels x 16 channels of an A operand fall on 64 different banks, with act(x*scale+shift) applied while
// staging) and the tile's dy ([voxel][16 co]); wave w owns rows 4w..4w+3 and keeps all 27 tap accumulators (108 registers).
// A dy operand is read once per 27 MFMAs.  Every wave writes its partial sums in the gradient's own layout to a slab of its own
// (no atomics; fixed-order fp64 reduce, k_slab_reduce_k), so the result is bit-reproducible.
// ================================================================================================================
namespace {

constexpr int W_BZ = 2, W_BY = 8, W_BX = 16, W_HZ = W_BZ + 2, W_HY = W_BY + 2, W_HX = W_BX + 2;
constexpr int W_HV = W_HZ * W_HY * W_HX, W_TV = W_BZ * W_BY * W_BX;     // 720 halo voxels, 256 tile voxels

struct WgradF32Args {
    ConvGeom g;
    SrcDesc s0, s1;
    const float* dy;
    float* slab;             // [gridDim.x * 4][27 * Cin * Cout]
    int64_t total;           // 27 * Cin * Cout
    int tz, ty, tx;
};

__global__ void __launch_bounds__(256) k_wgrad_f32_mfma(WgradF32Args a) {
    extern __shared__ float lds[];
    float* xs = lds;                       // [W_HV][16]
    float* ds = lds + W_HV * 16;           // [W_TV][16]
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, lx = lane & 15, lq = lane >> 4;
    const int ci0 = blockIdx.y * 16, co0 = blockIdx.z * 16;
    const bool second = ci0 >= a.s0.C;
    const float* sp = (const float*)(second ? a.s1.ptr : a.s0.ptr);
    const int sC = second ? a.s1.C : a.s0.C, cb = second ? ci0 - a.s0.C : ci0;
    const float* sc = second ? a.s1.scale : a.s0.scale;
    const float* sh = second ? a.s1.shift : a.s0.shift;
    const int act = second ? a.s1.act : a.s0.act;

    f32x4 acc[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int tiles = a.tz * a.ty * a.tx;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        int t = tile;
        const int bx = t % a.tx; t /= a.tx;
        const int by = t % a.ty, bz = t / a.ty;
        const int z0 = bz * W_BZ, y0 = by * W_BY, x0 = bx * W_BX;
        __syncthreads();   // the previous tile's MFMAs have read xs / ds
        for (int it = tid; it < W_HV * 4; it += 256) {
            const int hv = it >> 2, q = it & 3;
            const int hz = hv / (W_HY * W_HX), rem = hv - hz * (W_HY * W_HX), hy = rem / W_HX, hx = rem - hy * W_HX;
            const int iz = z0 + hz - 1, iy = y0 + hy - 1, ix = x0 + hx - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (iz >= 0 && iz < g.D && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) {
                const int c = cb + q * 4;
                v = *(const float4*)(sp + (((int64_t)iz * g.H + iy) * g.W + ix) * sC + c);
                if (sc) {
                    const float4 s4 = *(const float4*)(sc + c), h4 = *(const float4*)(sh + c);
                    v.x = v.x * s4.x + h4.x; v.y = v.y * s4.y + h4.y; v.z = v.z * s4.z + h4.z; v.w = v.w * s4.w + h4.w;
                }
                v.x = act_f(v.x, act); v.y = act_f(v.y, act); v.z = act_f(v.z, act); v.w = act_f(v.w, act);
            }
            *(float4*)(xs + hv * 16 + q * 4) = v;
        }
        for (int it = tid; it < W_TV * 4; it += 256) {
            const int tv = it >> 2, q = it & 3;
            const int tzz = tv / (W_BY * W_BX), rem = tv - tzz * (W_BY * W_BX), tyy = rem / W_BX, txx = rem - tyy * W_BX;
            const int oz = z0 + tzz, oy = y0 + tyy, ox = x0 + txx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (oz < g.Do && oy < g.Ho && ox < g.Wo)
                v = *(const float4*)(a.dy + (((int64_t)oz * g.Ho + oy) * g.Wo + ox) * g.Cout + co0 + q * 4);
            *(float4*)(ds + tv * 16 + q * 4) = v;
        }
        __syncthreads();
#pragma unroll 1
        for (int r = 0; r < 4; ++r) {
            const int row = wv * 4 + r, rz = row / W_BY, ry = row % W_BY;          // rows 0..15: (z, y) inside the tile
#pragma unroll 1
            for (int kk = 0; kk < 4; ++kk) {
                const int xv = kk * 4 + lq;                                         // this lane's voxel (k index) along x
                const float b = ds[((rz * W_BY + ry) * W_BX + xv) * 16 + lx];       // B[k = voxel][j = co]
                const float* xp = xs + ((rz * W_HY + ry) * W_HX + xv) * 16 + lx;    // A[i = ci][k = voxel] at tap (0,0,0)
#pragma unroll
                for (int t = 0; t < 27; ++t) {
                    const int kz = t / 9, ky = (t / 3) % 3, kx = t % 3;
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xp[((kz * W_HY + ky) * W_HX + kx) * 16], b, acc[t], 0, 0, 0);
                }
            }
        }
    }
    // lane (lq, lx) holds rows i = 4*lq + j (ci), column lx (co); torch layout index = (co*Cin + ci)*27 + tap
    float* sl = a.slab + (size_t)(blockIdx.x * 4 + wv) * a.total;
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) sl[((int64_t)(co0 + lx) * g.Cin + ci0 + 4 * lq + j) * 27 + t] = acc[t][j];
}

int wgrad_f32_blocks(const ConvGeom& g) {
    const int64_t tiles = (int64_t)((g.Do + W_BZ - 1) / W_BZ) * ((g.Ho + W_BY - 1) / W_BY) * ((g.Wo + W_BX - 1) / W_BX);
    const int64_t per_block = (int64_t)4 * 27 * g.Cin * g.Cout * 4;                // bytes of slabs one block column adds
    int64_t nb = ((int64_t)64 << 20) / per_block;
    if (nb < 1) nb = 1;
    if (nb > 256) nb = 256;
    return (int)(nb < tiles ? nb : tiles);
}

}  // namespace

bool wgrad_f32_mfma_supported(int dtype, const ConvGeom& g, const SrcDesc* src, int nsrc) {
    if (dtype != 0 || g.ks != 3 || g.stride != 1 || g.Cin % 16 || g.Cout % 16 || nsrc < 1 || nsrc > 2) return false;
    return src[0].C % 16 == 0 && (nsrc == 1 || src[1].C % 16 == 0);
}
size_t wgrad_f32_mfma_scratch_bytes(const ConvGeom& g) {
    int64_t So = (int64_t)g.Do * g.Ho * g.Wo;
    return (size_t)wgrad_f32_blocks(g) * 4 * 27 * g.Cin * g.Cout * 4 + bias_grad_scratch_bytes(g.Cout, So) + 512;
}
void launch_wgrad_f32_mfma(const ConvGeom& g, const SrcDesc* src, int nsrc, const float* dy, float* dw, float* db, void* scratch,
                           hipStream_t s) {
    WgradF32Args a;
    a.g = g; a.s0 = src[0]; a.s1 = nsrc > 1 ? src[1] : SrcDesc();
    if (nsrc == 1) a.s0.C = g.Cin;
    a.dy = dy; a.slab = (float*)scratch; a.total = (int64_t)27 * g.Cin * g.Cout;
    a.tz = (g.Do + W_BZ - 1) / W_BZ; a.ty = (g.Ho + W_BY - 1) / W_BY; a.tx = (g.Wo + W_BX - 1) / W_BX;
    const int nb = wgrad_f32_blocks(g);
    const size_t lds = (size_t)(W_HV * 16 + W_TV * 16) * sizeof(float);
    static std::atomic<uint64_t> once{0};
    set_max_lds_once(once, (const void*)k_wgrad_f32_mfma, (int)lds);
    k_wgrad_f32_mfma<<<dim3((unsigned)nb, (unsigned)(g.Cin / 16), (unsigned)(g.Cout / 16)), 256, lds, s>>>(a);
    slab_reduce_public(a.slab, nb * 4, a.total, dw, s);
    if (db) launch_bias_grad(0, dy, g.Cout, (int64_t)g.Do * g.Ho * g.Wo, db, (char*)scratch + (size_t)nb * 4 * a.total * 4 + 256, s);
}

}  // namespace unet
