// MFMA implicit-GEMM conv family for gfx950 (bf16 storage, fp32 accumulate).
//
// conv3d 3x3x3 stride 1 (Conv3d of unet.cpp:59-72; 98 % of the path's FLOPs) as an implicit GEMM on
// v_mfma_f32_16x16x32_bf16:   D[cout][voxel] += W[cout][k] * X[k][voxel],  k = (tap, cin).
//   * A operand  = filter tile, pre-packed in fragment order (one 16-B load per lane, L2-resident).
//   * B operand  = input patches, read with ds_read_b128 from an LDS halo tile
//                  [(BZ+2)][(BY+2)][(BX+2)][CK channels] that is staged once per CK-channel chunk and
//                  re-used by all 27 taps.
//   * the producer's norm + activation (scale/shift per channel, then relu/leaky/elu) is applied while
//     staging, zero padding is applied after it (padding pads the ACTIVATED tensor), and a channel
//     concat {skip, x} (unet.cpp:181) is just a second source pointer.
//   * epilogue: + bias, round to bf16, 8-B stores (4 consecutive channels per lane), optional per-block
//     {sum, sum of squares} per channel for the following norm (no extra pass over the tensor), optional
//     accumulate / two destinations (dgrad of a concat).
// The same kernel computes dgrad of a stride-1 conv: input = dL/dy, filter = flipped + transposed pack.
#include "device_util.h"

namespace unet {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct MfmaConvArgs {
    ConvGeom g;
    SrcDesc src[2];
    int nsrc;
    const void* w;       // packed filter
    const float* bias;   // nullptr: none
    void* out[2];        // channels-last bf16 destinations (split at outC[0] channels)
    int outC[2];
    int out_acc[2];
    int nout;
    float* stats;        // [nblk][Cout][2] or nullptr
    int tiles_x, tiles_y, tiles_z;
};

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    return (unsigned)__bfloat16_as_ushort(__float2bfloat16(lo)) | ((unsigned)__bfloat16_as_ushort(__float2bfloat16(hi)) << 16);
}
__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

// blocks that share an XCD (b % 8 equal) get a contiguous range of tiles: neighbouring tiles share halos in one L2
__device__ __forceinline__ int xcd_remap(int b, int n) {
    int q = n >> 3, r = n & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

template <int BZ, int BY, int BX, int CK, int NT>
__global__ void __launch_bounds__(256) k_mfma_conv3(MfmaConvArgs a) {
    constexpr int HZ = BZ + 2, HY = BY + 2, HX = BX + 2, NVOX = HZ * HY * HX;
    constexpr int G = CK / 8;                    // 16-B channel groups per voxel
    constexpr int VS = CK == 32 ? 96 : 32;       // LDS bytes per voxel (96: conflict-free ds_read_b128 for 64-B payloads)
    constexpr int TXM = BX < 16 ? BX : 16;       // m-tile = TYM rows x TXM columns of one z-plane
    constexpr int TYM = 16 / TXM;
    constexpr int MT = BZ * BY * BX / 16, MTW = MT / 4;
    constexpr int KSTEPS = CK == 32 ? 27 : 14;
    static_assert(MT % 4 == 0 && MTW >= 1, "tile must give every wave at least one m-tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, gq = lane >> 4;
    const int nblk = a.tiles_x * a.tiles_y * a.tiles_z;
    const int bid = xcd_remap(blockIdx.x, nblk);
    const int x0 = (bid % a.tiles_x) * BX, y0 = ((bid / a.tiles_x) % a.tiles_y) * BY, z0 = (bid / (a.tiles_x * a.tiles_y)) * BZ;
    const int nt0 = blockIdx.y * NT, NTT = g.Cout / 16;

    // this lane's voxel of each of the wave's m-tiles: tile-local coordinates and LDS byte offset
    int mz[MTW], my[MTW], mx[MTW], mbase[MTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        int mt = wave * MTW + i;
        constexpr int TPR = BX / TXM;            // m-tiles per row group (1 here: BX <= 16)
        constexpr int RG = BY / TYM;             // row groups per z-plane
        int zz = mt / (RG * TPR), rem = mt % (RG * TPR);
        mz[i] = zz; my[i] = (rem / TPR) * TYM + (j / TXM); mx[i] = (rem % TPR) * TXM + (j % TXM);
        mbase[i] = ((mz[i] * HY + my[i]) * HX + mx[i]) * VS + (CK == 32 ? gq : (gq & 1)) * 16;
    }

    f32x4 acc[MTW][NT];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int C0 = a.src[0].C;
    const int lg = tid % G;                      // this thread's channel group inside the chunk (256 % G == 0)
    const int nchunk = g.Cin / CK;
    const bf16x8* wp = (const bf16x8*)a.w;

    for (int q = 0; q < nchunk; ++q) {
        // ---- stage the halo tile of channels [q*CK, q*CK+CK) ----
        {
            int c = q * CK + lg * 8;
            int s = (a.nsrc > 1 && c >= C0) ? 1 : 0;
            const SrcDesc& sd = a.src[s];
            int cl = c - (s ? C0 : 0);
            const uint4* base = (const uint4*)((const char*)sd.ptr + (size_t)cl * 2);
            float sc[8], sh[8];
            const bool xf = sd.scale != nullptr;
            if (xf) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { sc[e] = sd.scale[cl + e]; sh[e] = sd.shift[cl + e]; }
            }
            const int act = sd.act;
            constexpr int UNITS = NVOX * G, ITERS = (UNITS + 255) / 256;
            __syncthreads();                      // previous chunk's reads are done
#pragma unroll 4
            for (int it = 0; it < ITERS; ++it) {
                int u = tid + it * 256;
                if (u < UNITS) {
                    int hv = u / G;
                    int hz = hv / (HY * HX), hr = hv % (HY * HX), hy = hr / HX, hx = hr % HX;
                    int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
                    uint4 v = make_uint4(0u, 0u, 0u, 0u);
                    if (gz >= 0 && gz < g.D && gy >= 0 && gy < g.H && gx >= 0 && gx < g.W) {
                        size_t vox = ((size_t)gz * g.H + gy) * g.W + gx;
                        v = *(const uint4*)((const char*)base + vox * (size_t)sd.C * 2);
                        if (xf || act) {
                            unsigned wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float lo = bf_lo(wv[e]), hi = bf_hi(wv[e]);
                                if (xf) { lo = fmaf(lo, sc[2 * e], sh[2 * e]); hi = fmaf(hi, sc[2 * e + 1], sh[2 * e + 1]); }
                                lo = act_f(lo, act); hi = act_f(hi, act);
                                wv[e] = pack_bf16x2(lo, hi);
                            }
                            v = make_uint4(wv[0], wv[1], wv[2], wv[3]);
                        }
                    }
                    *(uint4*)(smem + hv * VS + lg * 16) = v;
                }
            }
            __syncthreads();
        }
        // ---- 27 taps x MFMA ----
        const bf16x8* wq = wp + ((size_t)q * KSTEPS * NTT + nt0) * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            bf16x8 wf[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) wf[n] = wq[((size_t)ks * NTT + n) * 64];
            int toff;
            if (CK == 32) {
                const int kz = ks / 9, ky = (ks / 3) % 3, kx = ks % 3;
                toff = ((kz * HY + ky) * HX + kx) * VS;
            } else {
                const int t0 = 2 * ks, t1 = 2 * ks + 1 < 27 ? 2 * ks + 1 : 2 * ks;   // tap 27 does not exist: its filter is zero
                const int o0 = (((t0 / 9) * HY + (t0 / 3) % 3) * HX + t0 % 3) * VS;
                const int o1 = (((t1 / 9) * HY + (t1 / 3) % 3) * HX + t1 % 3) * VS;
                toff = (lane & 32) ? o1 : o0;
            }
#pragma unroll
            for (int i = 0; i < MTW; ++i) {
                bf16x8 xb = *(const bf16x8*)(smem + mbase[i] + toff);
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[n], xb, acc[i][n], 0, 0, 0);
            }
        }
    }

    // ---- epilogue ----
    float s1[NT][4], s2[NT][4];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[n][r] = 0.f; s2[n][r] = 0.f; }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        int c = (nt0 + n) * 16 + gq * 4;          // first of this lane's 4 output channels
        float b4[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r) b4[r] = a.bias[c + r];
        }
        int d = (a.nout > 1 && c >= a.outC[0]) ? 1 : 0;
        int cd = c - (d ? a.outC[0] : 0);
        char* obase = (char*)a.out[d];
        const int oC = a.outC[d], oacc = a.out_acc[d];
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
            int gz = z0 + mz[i], gy = y0 + my[i], gx = x0 + mx[i];
            if (gz < g.D && gy < g.H && gx < g.W && obase) {
                size_t vox = ((size_t)gz * g.H + gy) * g.W + gx;
                uint2* p = (uint2*)(obase + (vox * oC + cd) * 2);
                float v0 = acc[i][n][0] + b4[0], v1 = acc[i][n][1] + b4[1], v2 = acc[i][n][2] + b4[2], v3 = acc[i][n][3] + b4[3];
                if (oacc) {
                    uint2 old = *p;
                    v0 += bf_lo(old.x); v1 += bf_hi(old.x); v2 += bf_lo(old.y); v3 += bf_hi(old.y);
                }
                uint2 o;
                o.x = pack_bf16x2(v0, v1); o.y = pack_bf16x2(v2, v3);
                *p = o;
                if (a.stats) {   // statistics of the values as stored (rounded to bf16)
                    float r0 = bf_lo(o.x), r1 = bf_hi(o.x), r2 = bf_lo(o.y), r3 = bf_hi(o.y);
                    s1[n][0] += r0; s1[n][1] += r1; s1[n][2] += r2; s1[n][3] += r3;
                    s2[n][0] = fmaf(r0, r0, s2[n][0]); s2[n][1] = fmaf(r1, r1, s2[n][1]);
                    s2[n][2] = fmaf(r2, r2, s2[n][2]); s2[n][3] = fmaf(r3, r3, s2[n][3]);
                }
            }
        }
    }
    if (a.stats) {
        float* red = (float*)smem;                // [wave][NT*16][2]
        __syncthreads();                          // LDS tile no longer needed
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
            for (int w = 0; w < 4; ++w) { u += red[(w * NT * 16 + tid) * 2]; v += red[(w * NT * 16 + tid) * 2 + 1]; }
            int c = nt0 * 16 + tid;
            a.stats[((size_t)bid * g.Cout + c) * 2 + 0] = u;
            a.stats[((size_t)bid * g.Cout + c) * 2 + 1] = v;
        }
    }
}

// ---- filter packing: fp32 [Cout][Cin][27] -> bf16 fragments [chunk][kstep][ntile][lane][8] ----
// transposed_flip: the dgrad filter  W'[o = cin][i = cout][t] = W[i][o][26 - t]
__global__ void k_mfma_pack_w(const float* __restrict__ w, __bf16* __restrict__ out, int Ci, int Co, int CK, int flip, int CinOrig,
                              int CoutOrig) {
    int KSTEPS = CK == 32 ? 27 : 14, NTT = Co / 16;
    int64_t total = (int64_t)(Ci / CK) * KSTEPS * NTT * 64 * 8;
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    int e = (int)(idx & 7); int64_t r = idx >> 3;
    int lane = (int)(r & 63); r >>= 6;
    int nt = (int)(r % NTT); r /= NTT;
    int ks = (int)(r % KSTEPS); int q = (int)(r / KSTEPS);
    int o = nt * 16 + (lane & 15);
    int tap, i;
    if (CK == 32) { tap = ks; i = q * 32 + 8 * (lane >> 4) + e; }
    else { tap = 2 * ks + (lane >> 5); i = q * 16 + 8 * ((lane >> 4) & 1) + e; }
    float v = 0.f;
    if (tap < 27) v = flip ? w[((int64_t)i * CinOrig + o) * 27 + (26 - tap)] : w[((int64_t)o * CinOrig + i) * 27 + tap];
    (void)CoutOrig;
    out[idx] = (__bf16)v;
}

static inline int conv_ck(int Cin) { return Cin % 32 == 0 ? 32 : 16; }

bool mfma_conv_fwd_supported(int dtype, const ConvGeom& g, const SrcDesc* src, int nsrc) {
    if (dtype != 1 || g.ks != 3 || g.stride != 1) return false;
    if (g.Cin % 16 || g.Cout % 16) return false;
    for (int s = 0; s < nsrc; ++s)
        if (src[s].C % 16) return false;
    return true;
}
size_t mfma_conv_w_bytes(const ConvGeom& g) {
    int CK = conv_ck(g.Cin), KSTEPS = CK == 32 ? 27 : 14;
    return (size_t)(g.Cin / CK) * KSTEPS * (g.Cout / 16) * 64 * 16;
}
void launch_mfma_pack_conv_w(const float* w, void* w_fwd, void* w_dgrad, const ConvGeom& g, hipStream_t s) {
    if (w_fwd) {
        int CK = conv_ck(g.Cin);
        int64_t n = (int64_t)mfma_conv_w_bytes(g) / 2;
        k_mfma_pack_w<<<cdiv64(n, 256), 256, 0, s>>>(w, (__bf16*)w_fwd, g.Cin, g.Cout, CK, 0, g.Cin, g.Cout);
    }
    if (w_dgrad) {
        ConvGeom t = g;
        t.Cin = g.Cout; t.Cout = g.Cin;
        int CK = conv_ck(t.Cin);
        int64_t n = (int64_t)mfma_conv_w_bytes(t) / 2;
        k_mfma_pack_w<<<cdiv64(n, 256), 256, 0, s>>>(w, (__bf16*)w_dgrad, t.Cin, t.Cout, CK, 1, g.Cin, g.Cout);
    }
}

template <int BZ, int BY, int BX, int CK, int NT> static void launch_cfg(const MfmaConvArgs& a0, hipStream_t s) {
    MfmaConvArgs a = a0;
    a.tiles_x = (a.g.W + BX - 1) / BX; a.tiles_y = (a.g.H + BY - 1) / BY; a.tiles_z = (a.g.D + BZ - 1) / BZ;
    constexpr int VS = CK == 32 ? 96 : 32;
    constexpr size_t lds = (size_t)(BZ + 2) * (BY + 2) * (BX + 2) * VS;
    static_assert(lds >= 4 * NT * 16 * 2 * 4, "stats scratch must fit the tile buffer");
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)k_mfma_conv3<BZ, BY, BX, CK, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.tiles_z), (unsigned)(a.g.Cout / (16 * NT)));
    k_mfma_conv3<BZ, BY, BX, CK, NT><<<grid, 256, lds, s>>>(a);
}
template <int BZ, int BY, int BX, int CK> static void launch_nt(const MfmaConvArgs& a, hipStream_t s) {
    int ntt = a.g.Cout / 16;
    if (ntt % 4 == 0) launch_cfg<BZ, BY, BX, CK, 4>(a, s);
    else if (ntt % 2 == 0) launch_cfg<BZ, BY, BX, CK, 2>(a, s);
    else launch_cfg<BZ, BY, BX, CK, 1>(a, s);
}
int mfma_conv_blocks(const ConvGeom& g) {
    int CK = conv_ck(g.Cin);
    int bx = g.W >= 12 ? 16 : (g.W > 4 ? 8 : 4);
    int by = bx == 16 ? (CK == 32 ? 4 : 8) : (bx == 8 ? 8 : 4);
    return ((g.W + bx - 1) / bx) * ((g.H + by - 1) / by) * ((g.D + 3) / 4);
}
static void launch_mfma_conv_any(const MfmaConvArgs& a, hipStream_t s) {
    int CK = conv_ck(a.g.Cin);
    if (a.g.W >= 12) {
        if (CK == 32) launch_nt<4, 4, 16, 32>(a, s); else launch_nt<4, 8, 16, 16>(a, s);
    } else if (a.g.W > 4) {
        if (CK == 32) launch_nt<4, 8, 8, 32>(a, s); else launch_nt<4, 8, 8, 16>(a, s);
    } else {
        if (CK == 32) launch_nt<4, 4, 4, 32>(a, s); else launch_nt<4, 4, 4, 16>(a, s);
    }
}

void launch_mfma_conv_fwd(const ConvGeom& g, const SrcDesc* src, int nsrc, const void* w_mfma, const float* bias, void* out,
                          float* stats_partial, hipStream_t s) {
    MfmaConvArgs a;
    a.g = g; a.nsrc = nsrc; a.src[0] = src[0]; if (nsrc > 1) a.src[1] = src[1];
    a.w = w_mfma; a.bias = bias;
    a.out[0] = out; a.out[1] = nullptr; a.outC[0] = g.Cout; a.outC[1] = 0; a.out_acc[0] = 0; a.out_acc[1] = 0; a.nout = 1;
    a.stats = stats_partial;
    a.tiles_x = a.tiles_y = a.tiles_z = 0;
    launch_mfma_conv_any(a, s);
}

// dgrad of a stride-1 3x3x3 conv: conv of dL/dy [Cout ch] with the flipped filter into Cin channels,
// split over up to two destinations (the sources of a concat)
void launch_mfma_conv_dgrad(const ConvGeom& g, const void* dy, const void* w_mfma_dgrad, const DstGrad* dst, int ndst, hipStream_t s) {
    MfmaConvArgs a;
    a.g = g; a.g.Cin = g.Cout; a.g.Cout = g.Cin;   // roles swap; volume unchanged (stride 1, pad 1)
    a.g.D = g.Do; a.g.H = g.Ho; a.g.W = g.Wo; a.g.Do = g.D; a.g.Ho = g.H; a.g.Wo = g.W;
    a.nsrc = 1;
    a.src[0].ptr = dy; a.src[0].C = g.Cout;
    a.w = w_mfma_dgrad; a.bias = nullptr;
    for (int k = 0; k < 2; ++k) {
        a.out[k] = k < ndst ? dst[k].ptr : nullptr;
        a.outC[k] = k < ndst ? dst[k].C : 0;
        a.out_acc[k] = k < ndst ? dst[k].accumulate : 0;
    }
    a.nout = ndst;
    a.stats = nullptr;
    a.tiles_x = a.tiles_y = a.tiles_z = 0;
    launch_mfma_conv_any(a, s);
}

}  // namespace unet
