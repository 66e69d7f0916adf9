// Direct (non-MFMA) conv family: correct for every channel count and both element types; used for the
// fp32 parity configuration, for layers the MFMA kernels do not cover (Cin = 1, 6-channel heads, odd
// channel counts) and as the on-GPU cross-check of the MFMA kernels.  fp32 accumulation throughout.
// Semantics: Conv3d ks{1,3} stride{1,2} pad (ks-1)/2 (unet.cpp:59-72), ConvTranspose3d ks2 stride2
// (unet.cpp:46-57), and their autograd gradients (train.cpp:706).
#include "device_util.h"

namespace unet {

// ------------------------------------------------------------------------------------------------
// weight repacking
// ------------------------------------------------------------------------------------------------
__global__ void k_pack_conv_w(const float* __restrict__ w, float* __restrict__ wf, float* __restrict__ wd, int Cin, int Cout,
                              int k3, int CoutP, int CinP) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t nf = (int64_t)k3 * Cin * CoutP, nd = (int64_t)k3 * Cout * CinP;
    if (i < nf) {
        int co = (int)(i % CoutP); int64_t r = i / CoutP;
        int ci = (int)(r % Cin); int t = (int)(r / Cin);
        wf[i] = co < Cout ? w[((int64_t)co * Cin + ci) * k3 + t] : 0.f;
    }
    if (i < nd) {
        int ci = (int)(i % CinP); int64_t r = i / CinP;
        int co = (int)(r % Cout); int t = (int)(r / Cout);
        wd[i] = ci < Cin ? w[((int64_t)co * Cin + ci) * k3 + t] : 0.f;
    }
}
__global__ void k_pack_convt_w(const float* __restrict__ w, float* __restrict__ wf, float* __restrict__ wd, int Cin, int Cout,
                               int CoutP, int CinP) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t nf = (int64_t)8 * Cin * CoutP, nd = (int64_t)8 * Cout * CinP;
    if (i < nf) {
        int co = (int)(i % CoutP); int64_t r = i / CoutP;
        int ci = (int)(r % Cin); int t = (int)(r / Cin);
        wf[i] = co < Cout ? w[((int64_t)ci * Cout + co) * 8 + t] : 0.f;
    }
    if (i < nd) {
        int ci = (int)(i % CinP); int64_t r = i / CinP;
        int co = (int)(r % Cout); int t = (int)(r / Cout);
        wd[i] = ci < Cin ? w[((int64_t)ci * Cout + co) * 8 + t] : 0.f;
    }
}
void launch_pack_conv_w(const float* w, float* wf, float* wd, int Cin, int Cout, int k3, hipStream_t s) {
    int CoutP = round_up(Cout, 8), CinP = round_up(Cin, 8);
    int64_t na = (int64_t)k3 * Cin * CoutP, nb = (int64_t)k3 * Cout * CinP;
    int64_t n = na > nb ? na : nb;
    k_pack_conv_w<<<cdiv64(n, 256), 256, 0, s>>>(w, wf, wd, Cin, Cout, k3, CoutP, CinP);
}
void launch_pack_convt_w(const float* w, float* wf, float* wd, int Cin, int Cout, hipStream_t s) {
    int CoutP = round_up(Cout, 8), CinP = round_up(Cin, 8);
    int64_t a = (int64_t)8 * Cin * CoutP, b = (int64_t)8 * Cout * CinP;
    k_pack_convt_w<<<cdiv64(a > b ? a : b, 256), 256, 0, s>>>(w, wf, wd, Cin, Cout, CoutP, CinP);
}

// ------------------------------------------------------------------------------------------------
// conv forward: one thread = one output voxel x 8 output channels
// ------------------------------------------------------------------------------------------------
struct ConvFwdArgs {
    ConvGeom g;
    SrcDesc src[2];
    int nsrc;
    const float* w;   // [k3][Cin][CoutP]
    const float* bias;
    void* out;
    float* out_ncdhw;
};

template <typename T> __global__ void __launch_bounds__(256) k_conv_fwd_direct(ConvFwdArgs a) {
    const ConvGeom& g = a.g;
    int64_t So = (int64_t)g.Do * g.Ho * g.Wo;
    int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= So) return;
    int co0 = blockIdx.y * 8, CoutP = round_up(g.Cout, 8);
    int x = (int)(v % g.Wo); int64_t r = v / g.Wo;
    int y = (int)(r % g.Ho); int z = (int)(r / g.Ho);
    int pad = (g.ks - 1) / 2;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = (a.bias && co0 + j < g.Cout) ? a.bias[co0 + j] : 0.f;
    for (int kz = 0; kz < g.ks; ++kz) {
        int iz = z * g.stride + kz - pad;
        if (iz < 0 || iz >= g.D) continue;
        for (int ky = 0; ky < g.ks; ++ky) {
            int iy = y * g.stride + ky - pad;
            if (iy < 0 || iy >= g.H) continue;
            for (int kx = 0; kx < g.ks; ++kx) {
                int ix = x * g.stride + kx - pad;
                if (ix < 0 || ix >= g.W) continue;
                int64_t vin = ((int64_t)iz * g.H + iy) * g.W + ix;
                int tap = (kz * g.ks + ky) * g.ks + kx;
                int cb = 0;
                for (int s = 0; s < a.nsrc; ++s) {
                    const SrcDesc& sd = a.src[s];
                    for (int ci = 0; ci < sd.C; ++ci) {
                        float xv = view_ld<T>(sd, vin, ci);
                        const float* wr = a.w + ((int64_t)tap * g.Cin + cb + ci) * CoutP + co0;
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[j] = fmaf(xv, wr[j], acc[j]);
                    }
                    cb += sd.C;
                }
            }
        }
    }
    if (a.out_ncdhw) {
        for (int j = 0; j < 8; ++j)
            if (co0 + j < g.Cout) a.out_ncdhw[(int64_t)(co0 + j) * So + v] = acc[j];
    } else {
        T* o = (T*)a.out;
        for (int j = 0; j < 8; ++j)
            if (co0 + j < g.Cout) st<T>(o, v * g.Cout + co0 + j, acc[j]);
    }
}

// Cin = 1, 3x3x3 stride 1 (the network's first conv): one thread = one voxel x ALL output channels (<= 32), the 27
// neighbours are loaded once, the filter sits in LDS.  HBM-bound on the 2*Cout bytes per voxel it writes.
template <typename T, int CO> __global__ void __launch_bounds__(256) k_conv_first(ConvFwdArgs a) {
    const ConvGeom& g = a.g;
    const int CoutP = round_up(g.Cout, 8);   // packed filter [27][CoutP]: wave-uniform addresses -> scalar loads
    int64_t S = (int64_t)g.D * g.H * g.W;
    int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= S) return;
    int x = (int)(v % g.W); int64_t r = v / g.W;
    int y = (int)(r % g.H); int z = (int)(r / g.H);
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = (a.bias && c < g.Cout) ? a.bias[c] : 0.f;
#pragma unroll
    for (int t = 0; t < 27; ++t) {
        int iz = z + t / 9 - 1, iy = y + (t / 3) % 3 - 1, ix = x + t % 3 - 1;
        float xv = 0.f;
        if (iz >= 0 && iz < g.D && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) xv = view_ld<T>(a.src[0], ((int64_t)iz * g.H + iy) * g.W + ix, 0);
        const float* wr = a.w + (int64_t)t * CoutP;
#pragma unroll
        for (int c = 0; c < CO; ++c)
            if (c < CoutP) acc[c] = fmaf(xv, wr[c], acc[c]);
    }
    T* o = (T*)a.out + v * g.Cout;
    if (sizeof(T) == 2 && g.Cout % 8 == 0) {   // 16-B stores
#pragma unroll
        for (int c = 0; c < CO; c += 8) {
            if (c < g.Cout) {
                unsigned w4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    w4[e] = (unsigned)__bfloat16_as_ushort(__float2bfloat16(acc[c + 2 * e])) |
                            ((unsigned)__bfloat16_as_ushort(__float2bfloat16(acc[c + 2 * e + 1])) << 16);
                *(uint4*)((char*)o + c * 2) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
            }
        }
    } else {
        for (int c = 0; c < CO; ++c)
            if (c < g.Cout) st<T>(o, c, acc[c]);
    }
}

void launch_conv_fwd_direct(int dtype, const ConvGeom& g, const SrcDesc* src, int nsrc, const float* w, const float* bias,
                            void* out, float* out_ncdhw, hipStream_t s) {
    if (g.Cin == 1 && nsrc == 1 && g.ks == 3 && g.stride == 1 && g.Cout <= 32 && !out_ncdhw) {
        ConvFwdArgs f;
        f.g = g; f.nsrc = 1; f.src[0] = src[0]; f.w = w; f.bias = bias; f.out = out; f.out_ncdhw = nullptr;
        int64_t S = (int64_t)g.D * g.H * g.W;
        if (g.Cout <= 16) { UNET_DISPATCH(dtype, (k_conv_first<T, 16><<<cdiv64(S, 256), 256, 0, s>>>(f))); }
        else { UNET_DISPATCH(dtype, (k_conv_first<T, 32><<<cdiv64(S, 256), 256, 0, s>>>(f))); }
        return;
    }
    ConvFwdArgs a;
    a.g = g; a.nsrc = nsrc; a.src[0] = src[0]; if (nsrc > 1) a.src[1] = src[1];
    a.w = w; a.bias = bias; a.out = out; a.out_ncdhw = out_ncdhw;
    int64_t So = (int64_t)g.Do * g.Ho * g.Wo;
    dim3 grid(cdiv64(So, 256), (g.Cout + 7) / 8);
    UNET_DISPATCH(dtype, (k_conv_fwd_direct<T><<<grid, 256, 0, s>>>(a)));
}

// ------------------------------------------------------------------------------------------------
// conv dgrad: one thread = one input voxel x 8 input channels
// ------------------------------------------------------------------------------------------------
struct ConvDgradArgs {
    ConvGeom g;
    const void* dy;
    const float* w;  // [k3][Cout][CinP]
    DstGrad dst[2];
    int ndst;
};

template <typename T> __device__ __forceinline__ void write_grad(const DstGrad* dst, int ndst, int64_t v, int c, float val) {
    int cb = 0;
    for (int s = 0; s < ndst; ++s) {
        if (c < cb + dst[s].C) {
            if (dst[s].ptr) {
                T* p = (T*)dst[s].ptr;
                int64_t i = v * dst[s].C + (c - cb);
                if (dst[s].accumulate) val += ld<T>(p, i);
                st<T>(p, i, val);
            }
            return;
        }
        cb += dst[s].C;
    }
}

template <typename T> __global__ void __launch_bounds__(256) k_conv_dgrad_direct(ConvDgradArgs a) {
    const ConvGeom& g = a.g;
    int64_t S = (int64_t)g.D * g.H * g.W;
    int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= S) return;
    int ci0 = blockIdx.y * 8, CinP = round_up(g.Cin, 8);
    int ix = (int)(v % g.W); int64_t r = v / g.W;
    int iy = (int)(r % g.H); int iz = (int)(r / g.H);
    int pad = (g.ks - 1) / 2;
    const T* dy = (const T*)a.dy;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int kz = 0; kz < g.ks; ++kz) {
        int tz = iz + pad - kz;
        if (tz < 0 || (tz % g.stride) != 0) continue;
        int z = tz / g.stride;
        if (z >= g.Do) continue;
        for (int ky = 0; ky < g.ks; ++ky) {
            int ty = iy + pad - ky;
            if (ty < 0 || (ty % g.stride) != 0) continue;
            int y = ty / g.stride;
            if (y >= g.Ho) continue;
            for (int kx = 0; kx < g.ks; ++kx) {
                int tx = ix + pad - kx;
                if (tx < 0 || (tx % g.stride) != 0) continue;
                int x = tx / g.stride;
                if (x >= g.Wo) continue;
                int64_t vo = ((int64_t)z * g.Ho + y) * g.Wo + x;
                int tap = (kz * g.ks + ky) * g.ks + kx;
                for (int co = 0; co < g.Cout; ++co) {
                    float d = ld<T>(dy, vo * g.Cout + co);
                    const float* wr = a.w + ((int64_t)tap * g.Cout + co) * CinP + ci0;
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] = fmaf(d, wr[j], acc[j]);
                }
            }
        }
    }
    for (int j = 0; j < 8; ++j)
        if (ci0 + j < g.Cin) write_grad<T>(a.dst, a.ndst, v, ci0 + j, acc[j]);
}

void launch_conv_dgrad_direct(int dtype, const ConvGeom& g, const void* dy, const float* w, const DstGrad* dst, int ndst,
                              hipStream_t s) {
    ConvDgradArgs a;
    a.g = g; a.dy = dy; a.w = w; a.ndst = ndst; a.dst[0] = dst[0]; if (ndst > 1) a.dst[1] = dst[1];
    int64_t S = (int64_t)g.D * g.H * g.W;
    dim3 grid(cdiv64(S, 256), (g.Cin + 7) / 8);
    UNET_DISPATCH(dtype, (k_conv_dgrad_direct<T><<<grid, 256, 0, s>>>(a)));
}

// ------------------------------------------------------------------------------------------------
// wgrad (conv and conv_trans): one block = one (tap, ci); threads = output channels x row lanes
// ------------------------------------------------------------------------------------------------
struct WgradArgs {
    ConvGeom g;
    SrcDesc src[2];
    int nsrc;
    const void* dy;
    float* dw;
    int transposed;  // 0 conv, 1 conv_trans (ks2 stride2)
    int CW;          // threads along cout (power of two <= 256)
    float* slab;     // nullptr: add into dw directly; else partial sums slab[blockIdx.z][idx] (split over voxel rows)
    int64_t total;   // k3 * Cin * Cout
};

template <typename T> __global__ void __launch_bounds__(256) k_wgrad_direct(WgradArgs a) {
    const ConvGeom& g = a.g;
    __shared__ double red[256];
    int k3 = a.transposed ? 8 : g.ks * g.ks * g.ks;
    int tap = blockIdx.x % k3, ci = blockIdx.x / k3;
    int CW = a.CW, NV = 256 / CW;
    int co = blockIdx.y * CW + (threadIdx.x % CW), lane = threadIdx.x / CW;
    // which source holds input channel ci
    SrcDesc sd = a.src[0];
    int cl = ci;
    if (a.nsrc > 1 && ci >= a.src[0].C) { sd = a.src[1]; cl = ci - a.src[0].C; }
    const T* dy = (const T*)a.dy;
    double acc = 0.0;
    if (co < g.Cout) {
        if (!a.transposed) {
            int pad = (g.ks - 1) / 2;
            int kz = tap / (g.ks * g.ks), ky = (tap / g.ks) % g.ks, kx = tap % g.ks;
            const int rows = g.Do * g.Ho, rps = (rows + gridDim.z - 1) / gridDim.z;
            const int r0 = blockIdx.z * rps, r1 = r0 + rps < rows ? r0 + rps : rows;
            for (int row = r0 + lane; row < r1; row += NV) {
                int z = row / g.Ho, y = row % g.Ho;
                int iz = z * g.stride + kz - pad, iy = y * g.stride + ky - pad;
                if (iz < 0 || iz >= g.D || iy < 0 || iy >= g.H) continue;
                int64_t ibase = ((int64_t)iz * g.H + iy) * g.W, obase = (int64_t)row * g.Wo;
                float racc = 0.f;
                for (int x = 0; x < g.Wo; ++x) {
                    int ix = x * g.stride + kx - pad;
                    if (ix < 0 || ix >= g.W) continue;
                    racc = fmaf(view_ld<T>(sd, ibase + ix, cl), ld<T>(dy, (obase + x) * g.Cout + co), racc);
                }
                acc += (double)racc;
            }
        } else {
            int tz = tap >> 2, ty = (tap >> 1) & 1, tx = tap & 1;
            const int rows = g.D * g.H, rps = (rows + gridDim.z - 1) / gridDim.z;
            const int r0 = blockIdx.z * rps, r1 = r0 + rps < rows ? r0 + rps : rows;
            for (int row = r0 + lane; row < r1; row += NV) {
                int z = row / g.H, y = row % g.H;
                int64_t ibase = (int64_t)row * g.W, obase = ((int64_t)(2 * z + tz) * g.Ho + (2 * y + ty)) * g.Wo + tx;
                float racc = 0.f;
                for (int x = 0; x < g.W; ++x)
                    racc = fmaf(view_ld<T>(sd, ibase + x, cl), ld<T>(dy, (obase + 2 * x) * g.Cout + co), racc);
                acc += (double)racc;
            }
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (lane == 0 && co < g.Cout) {
        double s = 0.0;
        for (int l = 0; l < NV; ++l) s += red[l * CW + (threadIdx.x % CW)];
        int64_t idx = a.transposed ? ((int64_t)ci * g.Cout + co) * 8 + tap : ((int64_t)co * g.Cin + ci) * k3 + tap;
        if (a.slab) a.slab[(int64_t)blockIdx.z * a.total + idx] = (float)s;
        else a.dw[idx] += (float)s;
    }
}

// out[i] += sum_k slab[k][i]   (fixed order: deterministic)
// a block = LX consecutive outputs x 256/LX split lanes: coalesced slab reads, the lanes of an output share its split loop
template <int LX> __global__ void __launch_bounds__(256) k_slab_reduce_k(const float* __restrict__ slab, int nsplit, int64_t n,
                                                                         float* __restrict__ out, int64_t n1, float* __restrict__ out2) {
    constexpr int LY = 256 / LX;
    __shared__ double red[LY][LX];
    const int lx = threadIdx.x % LX, ly = threadIdx.x / LX;
    const int64_t i = (int64_t)blockIdx.x * LX + lx;
    double s = 0.0;
    if (i < n)
        for (int k = ly; k < nsplit; k += LY) s += slab[(int64_t)k * n + i];
    red[ly][lx] = s;
    __syncthreads();
    if (ly == 0 && i < n) {
        double tot = 0.0;
#pragma unroll
        for (int k = 0; k < LY; ++k) tot += red[k][lx];
        if (i < n1) out[i] += (float)tot;
        else if (out2) out2[i - n1] += (float)tot;
    }
}
// outputs [0,n1) go to out, [n1,n) to out2 (skipped when out2 is null)
static void slab_reduce2(const float* slab, int nsplit, int64_t n, float* out, int64_t n1, float* out2, hipStream_t s) {
    if (n <= 512 && nsplit >= 64) k_slab_reduce_k<4><<<cdiv64(n, 4), 256, 0, s>>>(slab, nsplit, n, out, n1, out2);
    else k_slab_reduce_k<32><<<cdiv64(n, 32), 256, 0, s>>>(slab, nsplit, n, out, n1, out2);
}
static void slab_reduce(const float* slab, int nsplit, int64_t n, float* out, hipStream_t s) { slab_reduce2(slab, nsplit, n, out, n, nullptr, s); }
void slab_reduce_public(const float* slab, int nsplit, int64_t n, float* out, hipStream_t s) { slab_reduce(slab, nsplit, n, out, s); }

// bias grad: db[c] += sum over voxels of dy[v][c]; one block per channel
template <typename T> __global__ void __launch_bounds__(256) k_bias_grad(const T* __restrict__ dy, int C, int64_t S, float* db, float* slab) {
    __shared__ double red[256];
    int c = blockIdx.x;
    double acc = 0.0;
    const int64_t per = (S + gridDim.y - 1) / gridDim.y, s0 = blockIdx.y * per, s1 = s0 + per < S ? s0 + per : S;
    for (int64_t v0 = s0 + threadIdx.x; v0 < s1; v0 += 256 * 64) {
        float r = 0.f;
        for (int k = 0; k < 64; ++k) {
            int64_t v = v0 + (int64_t)k * 256;
            if (v < s1) r += ld<T>(dy, v * C + c);
        }
        acc += (double)r;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (slab) slab[(int64_t)blockIdx.y * C + c] = (float)red[0];
        else db[c] += (float)red[0];
    }
}

static int wgrad_direct_split(const ConvGeom& g, int transposed) {
    int CW = 1;
    while (CW < g.Cout && CW < 256) CW <<= 1;
    int k3 = transposed ? 8 : g.ks * g.ks * g.ks;
    int64_t blocks = (int64_t)k3 * g.Cin * ((g.Cout + CW - 1) / CW);
    int rows = transposed ? g.D * g.H : g.Do * g.Ho;
    int NV = 256 / CW;
    int64_t want = 2048 / blocks;
    int maxs = (rows + NV - 1) / NV;      // at least one row per voxel lane
    if (want > maxs) want = maxs;
    return want < 1 ? 1 : (int)want;
}
static int bias_split(int64_t S) { int64_t n = S / 16384; return n < 1 ? 1 : (n > 256 ? 256 : (int)n); }
size_t wgrad_direct_scratch_bytes(const ConvGeom& g, int transposed) {
    int k3 = transposed ? 8 : g.ks * g.ks * g.ks;
    int64_t So = (int64_t)g.Do * g.Ho * g.Wo;
    return ((size_t)wgrad_direct_split(g, transposed) * k3 * g.Cin * g.Cout + (size_t)bias_split(So) * g.Cout) * 4 + 256;
}

static void launch_wgrad_common(int dtype, const ConvGeom& g, const SrcDesc* src, int nsrc, const void* dy, float* dw, float* db,
                                int transposed, void* scratch, hipStream_t s) {
    WgradArgs a;
    a.g = g; a.nsrc = nsrc; a.src[0] = src[0]; if (nsrc > 1) a.src[1] = src[1];
    a.dy = dy; a.dw = dw; a.transposed = transposed;
    int CW = 1;
    while (CW < g.Cout && CW < 256) CW <<= 1;
    a.CW = CW;
    int k3 = transposed ? 8 : g.ks * g.ks * g.ks;
    int nsplit = scratch ? wgrad_direct_split(g, transposed) : 1;
    a.total = (int64_t)k3 * g.Cin * g.Cout;
    a.slab = nsplit > 1 ? (float*)scratch : nullptr;
    dim3 grid((unsigned)(k3 * g.Cin), (unsigned)((g.Cout + CW - 1) / CW), (unsigned)nsplit);
    UNET_DISPATCH(dtype, (k_wgrad_direct<T><<<grid, 256, 0, s>>>(a)));
    if (nsplit > 1) slab_reduce(a.slab, nsplit, a.total, dw, s);
    if (db) {
        int64_t So = (int64_t)g.Do * g.Ho * g.Wo;
        int bs = scratch ? bias_split(So) : 1;
        float* bslab = bs > 1 ? (float*)scratch + (size_t)nsplit * a.total : nullptr;
        UNET_DISPATCH(dtype, (k_bias_grad<T><<<dim3(g.Cout, bs), 256, 0, s>>>((const T*)dy, g.Cout, So, db, bslab)));
        if (bs > 1) slab_reduce(bslab, bs, g.Cout, db, s);
    }
}
// db[c] += sum_v dy[v][c]   (scratch: bias_grad_scratch_bytes, or nullptr for a single-block-per-channel pass)
size_t bias_grad_scratch_bytes(int C, int64_t S) {
    size_t a = (size_t)bias_split(S) * C * 4, b = (size_t)stats_blocks(S) * C * 4;
    return (a > b ? a : b) + 256;
}
void launch_bias_grad(int dtype, const void* dy, int C, int64_t S, float* db, void* scratch, hipStream_t s) {
    if (scratch) {   // coalesced 16-B loads when the channel count allows
        int nb = launch_colsum_partial8(dtype, dy, C, S, (float*)scratch, s);
        if (nb > 0) { slab_reduce((const float*)scratch, nb, C, db, s); return; }
    }
    int bs = scratch ? bias_split(S) : 1;
    float* bslab = bs > 1 ? (float*)scratch : nullptr;
    UNET_DISPATCH(dtype, (k_bias_grad<T><<<dim3(C, bs), 256, 0, s>>>((const T*)dy, C, S, db, bslab)));
    if (bs > 1) slab_reduce(bslab, bs, C, db, s);
}

void launch_conv_wgrad_direct(int dtype, const ConvGeom& g, const SrcDesc* src, int nsrc, const void* dy, float* dw, float* db,
                              void* scratch, hipStream_t s) {
    launch_wgrad_common(dtype, g, src, nsrc, dy, dw, db, 0, scratch, s);
}
void launch_convt_wgrad_direct(int dtype, const ConvGeom& g, const SrcDesc* src, int nsrc, const void* dy, float* dw, float* db,
                               void* scratch, hipStream_t s) {
    launch_wgrad_common(dtype, g, src, nsrc, dy, dw, db, 1, scratch, s);
}

// ------------------------------------------------------------------------------------------------
// conv_trans ks2 stride2 forward / dgrad
// ------------------------------------------------------------------------------------------------
template <typename T> __global__ void __launch_bounds__(256) k_convt_fwd_direct(ConvFwdArgs a) {
    const ConvGeom& g = a.g;
    int64_t So = (int64_t)g.Do * g.Ho * g.Wo;
    int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= So) return;
    int co0 = blockIdx.y * 8, CoutP = round_up(g.Cout, 8);
    int x = (int)(v % g.Wo); int64_t r = v / g.Wo;
    int y = (int)(r % g.Ho); int z = (int)(r / g.Ho);
    int tap = ((z & 1) * 2 + (y & 1)) * 2 + (x & 1);
    int64_t vin = ((int64_t)(z >> 1) * g.H + (y >> 1)) * g.W + (x >> 1);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = (a.bias && co0 + j < g.Cout) ? a.bias[co0 + j] : 0.f;
    int cb = 0;
    for (int s = 0; s < a.nsrc; ++s) {
        const SrcDesc& sd = a.src[s];
        for (int ci = 0; ci < sd.C; ++ci) {
            float xv = view_ld<T>(sd, vin, ci);
            const float* wr = a.w + ((int64_t)tap * g.Cin + cb + ci) * CoutP + co0;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(xv, wr[j], acc[j]);
        }
        cb += sd.C;
    }
    T* o = (T*)a.out;
    for (int j = 0; j < 8; ++j)
        if (co0 + j < g.Cout) st<T>(o, v * g.Cout + co0 + j, acc[j]);
}
void launch_convt_fwd_direct(int dtype, const ConvGeom& g, const SrcDesc* src, int nsrc, const float* w, const float* bias,
                             void* out, hipStream_t s) {
    ConvFwdArgs a;
    a.g = g; a.nsrc = nsrc; a.src[0] = src[0]; if (nsrc > 1) a.src[1] = src[1];
    a.w = w; a.bias = bias; a.out = out; a.out_ncdhw = nullptr;
    int64_t So = (int64_t)g.Do * g.Ho * g.Wo;
    dim3 grid(cdiv64(So, 256), (g.Cout + 7) / 8);
    UNET_DISPATCH(dtype, (k_convt_fwd_direct<T><<<grid, 256, 0, s>>>(a)));
}

template <typename T> __global__ void __launch_bounds__(256) k_convt_dgrad_direct(ConvDgradArgs a) {
    const ConvGeom& g = a.g;
    int64_t S = (int64_t)g.D * g.H * g.W;
    int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= S) return;
    int ci0 = blockIdx.y * 8, CinP = round_up(g.Cin, 8);
    int x = (int)(v % g.W); int64_t r = v / g.W;
    int y = (int)(r % g.H); int z = (int)(r / g.H);
    const T* dy = (const T*)a.dy;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int tap = 0; tap < 8; ++tap) {
        int64_t vo = ((int64_t)(2 * z + (tap >> 2)) * g.Ho + (2 * y + ((tap >> 1) & 1))) * g.Wo + (2 * x + (tap & 1));
        for (int co = 0; co < g.Cout; ++co) {
            float d = ld<T>(dy, vo * g.Cout + co);
            const float* wr = a.w + ((int64_t)tap * g.Cout + co) * CinP + ci0;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(d, wr[j], acc[j]);
        }
    }
    for (int j = 0; j < 8; ++j)
        if (ci0 + j < g.Cin) write_grad<T>(a.dst, a.ndst, v, ci0 + j, acc[j]);
}
void launch_convt_dgrad_direct(int dtype, const ConvGeom& g, const void* dy, const float* w, const DstGrad* dst, int ndst,
                               hipStream_t s) {
    ConvDgradArgs a;
    a.g = g; a.dy = dy; a.w = w; a.ndst = ndst; a.dst[0] = dst[0]; if (ndst > 1) a.dst[1] = dst[1];
    int64_t S = (int64_t)g.D * g.H * g.W;
    dim3 grid(cdiv64(S, 256), (g.Cin + 7) / 8);
    UNET_DISPATCH(dtype, (k_convt_dgrad_direct<T><<<grid, 256, 0, s>>>(a)));
}

// ------------------------------------------------------------------------------------------------
// wgrad for layers with few weights (k3*Cin*Cout <= 1024: the Cin = 1 first conv, the 6-channel 1x1 heads):
// every block owns a strided set of output rows, stages the row's input neighbourhood and dy in LDS (fp32, input
// already transformed) and every thread accumulates up to 4 weight gradients over the row.  Per-block partials go
// to a slab that k_slab_reduce sums in a fixed order.  HBM-bound: reads dy and the input once.
// ------------------------------------------------------------------------------------------------
struct WgradSmallArgs {
    ConvGeom g;
    SrcDesc src[2];
    int nsrc;
    const void* dy;
    float* slab;   // [gridDim.x][k3*Cin*Cout]
};

template <typename T> __global__ void __launch_bounds__(256) k_wgrad_small(WgradSmallArgs a) {
    extern __shared__ float sm[];
    const ConvGeom& g = a.g;
    const int ks = g.ks, k3 = ks * ks * ks, pad = (ks - 1) / 2;
    const int O = k3 * g.Cin * g.Cout;
    const int WI = (g.Wo - 1) * g.stride + ks;      // input columns a row of outputs touches
    float* sa = sm;                                  // [ks*ks rows][WI][Cin]
    float* sd = sm + ks * ks * WI * g.Cin;           // [Wo][Cout]
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int oc[4], oci[4], orow[4], okx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int o = threadIdx.x + k * 256;
        if (o >= O) o = 0;
        int tap = o % k3, r = o / k3;                // torch layout index = (co*Cin + ci)*k3 + tap
        oci[k] = r % g.Cin; oc[k] = r / g.Cin;
        orow[k] = tap / ks; okx[k] = tap % ks;       // orow = kz*ks + ky
    }
    const T* dy = (const T*)a.dy;
    const int C0 = a.src[0].C;
    for (int row = blockIdx.x; row < g.Do * g.Ho; row += gridDim.x) {
        const int z = row / g.Ho, y = row % g.Ho;
        __syncthreads();
        for (int i = threadIdx.x; i < ks * ks * WI * g.Cin; i += 256) {
            int ci = i % g.Cin, r = i / g.Cin;
            int xi = r % WI, rr = r / WI;
            int iz = z * g.stride + rr / ks - pad, iy = y * g.stride + rr % ks - pad, ix = xi - pad;
            float v = 0.f;
            if (iz >= 0 && iz < g.D && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) {
                int64_t vox = ((int64_t)iz * g.H + iy) * g.W + ix;
                v = (a.nsrc > 1 && ci >= C0) ? view_ld<T>(a.src[1], vox, ci - C0) : view_ld<T>(a.src[0], vox, ci);
            }
            sa[i] = v;
        }
        for (int i = threadIdx.x; i < g.Wo * g.Cout; i += 256) sd[i] = ld<T>(dy, (int64_t)row * g.Wo * g.Cout + i);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (threadIdx.x + k * 256 < O) {
                const float* pa = sa + ((orow[k] * WI + okx[k]) * g.Cin + oci[k]);
                const float* pd = sd + oc[k];
                float r = 0.f;
                for (int x = 0; x < g.Wo; ++x) r = fmaf(pa[x * g.stride * g.Cin], pd[x * g.Cout], r);
                acc[k] += r;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int o = threadIdx.x + k * 256;
        if (o < O) a.slab[(int64_t)blockIdx.x * O + o] = acc[k];
    }
}

// ---- register-accumulating small wgrads -----------------------------------------------------------------------------
// Few weights, many voxels (the 1x1x1 heads, the Cin=1 first conv): a thread owns a slice of the weights (a "chunk") and keeps
// its accumulators in registers while it streams voxels; lanes that own the same chunk are summed with xor-shuffles, waves
// through LDS, blocks through the slab (fixed order everywhere).
template <int NACC, typename F>
__device__ __forceinline__ void small_wgrad_epilogue(float (&acc)[NACC], int nchunk, float* sm, float* slab_row, F omap) {
    // Through LDS, a third of the accumulators at a time: every lane writes its values as [value][lane], a lane then sums one (value,
    // chunk) over the lanes that own that chunk, the four wave results are added in wave order.  (The first version ran a xor-shuffle
    // butterfly per accumulator: 6 x 102 ds_bpermute per wave, ~8 us per CU whatever the volume -- the whole cost of the head backward
    // on the small levels.)  LDS: small_wgrad_lds_bytes(NACC, nchunk).
    constexpr int VP = (NACC + 2) / 3, PAD = 65;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* smw = sm + wave * VP * PAD;
    float* smx = sm + 4 * VP * PAD;
    const int nout = VP * nchunk;
#pragma unroll
    for (int ps = 0; ps < 3; ++ps) {
#pragma unroll
        for (int a = 0; a < VP; ++a)
            if (ps * VP + a < NACC) smw[a * PAD + lane] = acc[ps * VP + a];
        __syncthreads();
        for (int o = lane; o < nout; o += 64) {
            const int a = o / nchunk, ch = o % nchunk;
            float r = 0.f;
            for (int l = ch; l < 64; l += nchunk) r += smw[a * PAD + l];
            smx[wave * nout + o] = r;
        }
        __syncthreads();
        for (int o = threadIdx.x; o < nout; o += 256) {
            const int a = o / nchunk, ch = o % nchunk, i = ps * VP + a;
            if (i < NACC) {
                const float t = smx[o] + smx[nout + o] + smx[2 * nout + o] + smx[3 * nout + o];
                const int dst = omap(ch, i);
                if (dst >= 0) slab_row[dst] = t;
            }
        }
        __syncthreads();
    }
}
static inline size_t small_wgrad_lds_bytes(int nacc, int nchunk) { const int vp = (nacc + 2) / 3; return (size_t)(4 * vp * 65 + 4 * vp * nchunk) * 4; }

struct WgradRegArgs {
    ConvGeom g;
    SrcDesc src;
    const void* dy;
    float* slab;      // [gridDim.x][O + Cout]  (bias sums appended)
    int lc;           // log2(chunks per voxel)
};

// 1x1x1 conv, Cout <= CO <= 8, Cin = 16 * 2^lc: a thread owns 16 input channels x all outputs
template <typename T, int CO> __global__ void __launch_bounds__(256) k_wgrad_head(WgradRegArgs a) {
    extern __shared__ float sm[];
    constexpr int NACC = CO * 16 + CO;
    const ConvGeom& g = a.g;
    const int nchunk = 1 << a.lc, chunk = threadIdx.x & (nchunk - 1), c0 = chunk * 16;
    const int64_t S = (int64_t)g.D * g.H * g.W;
    const int64_t vstride = ((int64_t)gridDim.x * 256) >> a.lc;
    const T* dy = (const T*)a.dy;
    const bool plain = !a.src.scale && a.src.act == 0;
    float acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.f;
    for (int64_t v0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> a.lc; v0 < S; v0 += 4 * vstride) {
        float xs[4][16], ds[4][CO];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t v = v0 + u * vstride;
            const bool ok = v < S;
            if constexpr (sizeof(T) == 2) {
                if (plain) {
                    uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0;
                    if (ok) {
                        const uint4* q = (const uint4*)((const bf16*)a.src.ptr + v * a.src.C + c0);
                        r0 = q[0]; r1 = q[1];
                    }
                    const unsigned w[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
                    for (int k = 0; k < 8; ++k) { xs[u][2 * k] = __uint_as_float(w[k] << 16); xs[u][2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u); }
                } else {
#pragma unroll
                    for (int k = 0; k < 16; ++k) xs[u][k] = ok ? view_ld<T>(a.src, v, c0 + k) : 0.f;
                }
            } else {
#pragma unroll
                for (int k = 0; k < 16; ++k) xs[u][k] = ok ? view_ld<T>(a.src, v, c0 + k) : 0.f;
            }
#pragma unroll
            for (int c = 0; c < CO; ++c) ds[u][c] = (ok && c < g.Cout) ? ld<T>(dy, v * g.Cout + c) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int c = 0; c < CO; ++c) {
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[c * 16 + k] = fmaf(ds[u][c], xs[u][k], acc[c * 16 + k]);
                acc[CO * 16 + c] += ds[u][c];
            }
        }
    }
    const int Cin = g.Cin, Cout = g.Cout, O = Cin * Cout;
    small_wgrad_epilogue<NACC>(acc, nchunk, sm, a.slab + (int64_t)blockIdx.x * (O + Cout), [=](int ch, int i) {
        if (i < CO * 16) { int c = i / 16, k = i % 16; return c < Cout ? c * Cin + ch * 16 + k : -1; }
        int c = i - CO * 16;
        return (ch == 0 && c < Cout) ? O + c : -1;
    });
}

// 3x3x3 stride-1 conv with Cin = 1, Cout = 4 * 2^lc: a thread owns 4 outputs x 27 taps and walks runs of 4 voxels along x.
// A block takes tiles of FIRST_TY output rows (all of x); the 3 x (FIRST_TY+2) input rows a tile touches are transformed once
// into LDS as fp32 (zero halo), so a run's 3x3x6 window is nine 16-B + 8-B LDS reads.
constexpr int FIRST_TY = 8;
template <typename T> __global__ void __launch_bounds__(256) k_wgrad_first(WgradRegArgs a) {
    extern __shared__ float sm[];
    constexpr int NACC = 27 * 4 + 4;
    const ConvGeom& g = a.g;
    const int nchunk = 1 << a.lc, chunk = threadIdx.x & (nchunk - 1), co0 = chunk * 4;
    const int W4 = g.W >> 2, WP = (g.W + 2 + 3) & ~3;       // padded row pitch (floats), 16-B aligned runs
    const int ytiles = (g.H + FIRST_TY - 1) / FIRST_TY, tiles = g.D * ytiles;
    const int nin = 3 * (FIRST_TY + 2) * WP;
    const int items = (FIRST_TY * W4) << a.lc;
    const T* dy = (const T*)a.dy;
    float acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.f;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int z = tile / ytiles, y0 = (tile % ytiles) * FIRST_TY;
        __syncthreads();
        for (int i = threadIdx.x; i < nin; i += 256) {
            const int px = i % WP, r = i / WP, ry = r % (FIRST_TY + 2), rz = r / (FIRST_TY + 2);
            const int iz = z + rz - 1, iy = y0 + ry - 1, ix = px - 1;
            float v = 0.f;
            if (iz >= 0 && iz < g.D && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) v = view_ld<T>(a.src, ((int64_t)iz * g.H + iy) * g.W + ix, 0);
            sm[i] = v;
        }
        __syncthreads();
        for (int it = threadIdx.x; it < items; it += 256) {
            const int run = it >> a.lc, ty = run / W4, x0 = (run % W4) * 4;
            if (y0 + ty >= g.H) continue;
            float in[3][3][6], ds[4][4];
#pragma unroll
            for (int kz = 0; kz < 3; ++kz)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const float* row = sm + (kz * (FIRST_TY + 2) + ty + ky) * WP + x0;
                    const float4 q = *(const float4*)row;
                    const float2 q2 = *(const float2*)(row + 4);
                    in[kz][ky][0] = q.x; in[kz][ky][1] = q.y; in[kz][ky][2] = q.z; in[kz][ky][3] = q.w;
                    in[kz][ky][4] = q2.x; in[kz][ky][5] = q2.y;
                }
            const int64_t vox = ((int64_t)z * g.H + y0 + ty) * g.W + x0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if constexpr (sizeof(T) == 2) {
                    const uint2 r = *(const uint2*)((const bf16*)dy + (vox + u) * g.Cout + co0);
                    ds[u][0] = __uint_as_float(r.x << 16); ds[u][1] = __uint_as_float(r.x & 0xffff0000u);
                    ds[u][2] = __uint_as_float(r.y << 16); ds[u][3] = __uint_as_float(r.y & 0xffff0000u);
                } else {
                    const float4 r = *(const float4*)((const float*)dy + (vox + u) * g.Cout + co0);
                    ds[u][0] = r.x; ds[u][1] = r.y; ds[u][2] = r.z; ds[u][3] = r.w;
                }
            }
#pragma unroll
            for (int kz = 0; kz < 3; ++kz)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int u = 0; u < 4; ++u)
#pragma unroll
                            for (int c = 0; c < 4; ++c)
                                acc[((kz * 3 + ky) * 3 + kx) * 4 + c] = fmaf(in[kz][ky][u + kx], ds[u][c], acc[((kz * 3 + ky) * 3 + kx) * 4 + c]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[108 + c] += ds[u][c];
        }
    }
    __syncthreads();
    const int Cout = g.Cout, O = 27 * Cout;
    small_wgrad_epilogue<NACC>(acc, nchunk, sm, a.slab + (int64_t)blockIdx.x * (O + Cout), [=](int ch, int i) {
        if (i < 108) return (ch * 4 + i % 4) * 27 + i / 4;      // torch layout [co][ci=0][tap]
        return O + ch * 4 + (i - 108);
    });
}

// ---- the 1x1x1 heads (Conv3d(C, out_count, 1), unet.cpp:157-160), forward and fused backward ----------------------------------
// Lanes = (voxel, 16-channel chunk) as in k_wgrad_head; the filter ([Cout][Cin] fp32, torch layout, no pack) sits in LDS.
// ss: LDS image {scale[Cin], shift[Cin]} of a viewed bf16 source (nullptr: plain source, or the scalar path below).  A viewed source
// is the raw tensor of a norm layer whose activated copy was never written (the head is its only reader: engine.cpp, layout()):
// two 16-B loads + the transform of view_ld (v * scale + shift, then the activation) in registers.
template <typename T> __device__ __forceinline__ void load_chunk16(const SrcDesc& src, bool plain, int64_t v, int c0, bool ok, float (&x)[16],
                                                                  const float* ss = nullptr, int Cin = 0) {
    if constexpr (sizeof(T) == 2) {
        if (plain || ss) {
            uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0;
            if (ok) {
                const uint4* q = (const uint4*)((const bf16*)src.ptr + v * src.C + c0);
                r0 = q[0]; r1 = q[1];
            }
            const unsigned w[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) { x[2 * k] = __uint_as_float(w[k] << 16); x[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u); }
            if (!plain) {
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    const float4 sc = *(const float4*)(ss + c0 + 4 * k4), sh = *(const float4*)(ss + Cin + c0 + 4 * k4);
                    x[4 * k4] = act_f(x[4 * k4] * sc.x + sh.x, src.act); x[4 * k4 + 1] = act_f(x[4 * k4 + 1] * sc.y + sh.y, src.act);
                    x[4 * k4 + 2] = act_f(x[4 * k4 + 2] * sc.z + sh.z, src.act); x[4 * k4 + 3] = act_f(x[4 * k4 + 3] * sc.w + sh.w, src.act);
                }
                if (!ok) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) x[k] = 0.f;
                }
            }
            return;
        }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = ok ? view_ld<T>(src, v, c0 + k) : 0.f;
}
// the viewed bf16 source's {scale, shift} behind the filter in LDS (head kernels; the caller synchronizes)
template <typename T> __device__ __forceinline__ const float* stage_view(const SrcDesc& src, float* sm_after_filter, int Cin) {
    if (sizeof(T) != 2 || !src.scale) return nullptr;
    for (int i = threadIdx.x; i < Cin; i += 256) { sm_after_filter[i] = src.scale[i]; sm_after_filter[Cin + i] = src.shift[i]; }
    return sm_after_filter;
}

struct HeadArgs {
    SrcDesc src;
    const float* w;        // [Cout][Cin]
    const float* bias;     // [Cout] or nullptr
    int Cin, Cout, lc;     // lc = log2(Cin / 16)
    int64_t S;
    // forward
    void* y;               // channels-last [S][Cout] in T, or nullptr
    float* out;            // fp32 NCDHW [Cout][S] (results[level]), or nullptr
    // backward
    const float* dy_planes;  // fp32 NCDHW gradient of results[level] ...
    const void* dy_cl;       // ... or channels-last [S][Cout] in T
    DstGrad dst;             // dL/d(source view): written or accumulated; ptr may be null
    float* slab;             // [gridDim.x][Cout*Cin + Cout] wgrad + bias partials (nullptr: no parameter gradients)
};

template <typename T, int CO> __global__ void __launch_bounds__(256) k_head_fwd(HeadArgs a) {
    extern __shared__ float sm[];      // [CO][Cin] (rows >= Cout zero), then {scale, shift}[Cin] of a viewed source
    for (int i = threadIdx.x; i < CO * a.Cin; i += 256) sm[i] = i < a.Cout * a.Cin ? a.w[i] : 0.f;
    const float* ss = stage_view<T>(a.src, sm + CO * a.Cin, a.Cin);
    __syncthreads();
    const int nchunk = 1 << a.lc, chunk = threadIdx.x & (nchunk - 1), c0 = chunk * 16;
    const int64_t vstride = ((int64_t)gridDim.x * 256) >> a.lc;
    const bool plain = !a.src.scale && a.src.act == 0;
    float b[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) b[c] = (a.bias && c < a.Cout) ? a.bias[c] : 0.f;
    // the grid covers whole waves of (voxel, chunk) items: every lane takes part in the shuffles
    for (int64_t v0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> a.lc; v0 < ((a.S + 63) & ~(int64_t)63); v0 += vstride) {
        const bool ok = v0 < a.S;
        float x[16], o[CO];
        load_chunk16<T>(a.src, plain, v0, c0, ok, x, ss, a.Cin);
#pragma unroll
        for (int c = 0; c < CO; ++c) {
            const float4* wr = (const float4*)(sm + c * a.Cin + c0);
            float r = 0.f;
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const float4 w4 = wr[k4];
                r = fmaf(x[4 * k4], w4.x, r); r = fmaf(x[4 * k4 + 1], w4.y, r); r = fmaf(x[4 * k4 + 2], w4.z, r); r = fmaf(x[4 * k4 + 3], w4.w, r);
            }
            for (int m = 1; m < nchunk; m <<= 1) r += __shfl_xor(r, m);
            o[c] = r + b[c];
        }
        if (ok && chunk == 0) {
#pragma unroll
            for (int c = 0; c < CO; ++c) {
                if (c < a.Cout) {
                    if (a.out) a.out[(int64_t)c * a.S + v0] = o[c];
                    if (a.y) st<T>((T*)a.y, v0 * a.Cout + c, o[c]);
                }
            }
        }
    }
}

// dL/dW, dL/db and dL/d(source) of a head in one pass over (source, dy): replaces gradient import + wgrad + dgrad
template <typename T, int CO> __global__ void __launch_bounds__(256) k_head_bwd(HeadArgs a) {
    extern __shared__ float sm[];      // [CO][Cin] filter (+ {scale, shift}[Cin] of a viewed source), later the wgrad reduction scratch
    constexpr int NACC = CO * 16 + CO;
    for (int i = threadIdx.x; i < CO * a.Cin; i += 256) sm[i] = i < a.Cout * a.Cin ? a.w[i] : 0.f;
    const float* ss = stage_view<T>(a.src, sm + CO * a.Cin, a.Cin);
    __syncthreads();
    const int nchunk = 1 << a.lc, chunk = threadIdx.x & (nchunk - 1), c0 = chunk * 16;
    const int64_t vstride = ((int64_t)gridDim.x * 256) >> a.lc;
    const bool plain = !a.src.scale && a.src.act == 0;
    float acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.f;
    // the next voxel's operands are requested before this voxel's 200 FMAs (a thread walks ~16 voxels; one load -> use round trip
    // per voxel left the level-0 head at 63 us for 184 MB)
    auto fetch = [&](int64_t v, float (&xo)[16], float (&dout)[CO]) {
        load_chunk16<T>(a.src, plain, v, c0, true, xo, ss, a.Cin);
#pragma unroll
        for (int c = 0; c < CO; ++c) {
            dout[c] = 0.f;
            if (c < a.Cout) dout[c] = a.dy_planes ? a.dy_planes[(int64_t)c * a.S + v] : ld<T>((const T*)a.dy_cl, v * a.Cout + c);
        }
    };
    float xn[16], dn[CO];
    int64_t v0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> a.lc;
    if (v0 < a.S) fetch(v0, xn, dn);
    for (; v0 < a.S; v0 += vstride) {
        float x[16], d[CO];
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = xn[k];
#pragma unroll
        for (int c = 0; c < CO; ++c) d[c] = dn[c];
        if (v0 + vstride < a.S) fetch(v0 + vstride, xn, dn);
        if (a.slab) {
#pragma unroll
            for (int c = 0; c < CO; ++c) {
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[c * 16 + k] = fmaf(d[c], x[k], acc[c * 16 + k]);
                acc[CO * 16 + c] += d[c];
            }
        }
        if (a.dst.ptr) {
            float g[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) g[k] = 0.f;
#pragma unroll
            for (int c = 0; c < CO; ++c) {
                const float4* wr = (const float4*)(sm + c * a.Cin + c0);
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    const float4 w4 = wr[k4];
                    g[4 * k4] = fmaf(d[c], w4.x, g[4 * k4]); g[4 * k4 + 1] = fmaf(d[c], w4.y, g[4 * k4 + 1]);
                    g[4 * k4 + 2] = fmaf(d[c], w4.z, g[4 * k4 + 2]); g[4 * k4 + 3] = fmaf(d[c], w4.w, g[4 * k4 + 3]);
                }
            }
            if constexpr (sizeof(T) == 2) {
                uint4* q = (uint4*)((bf16*)a.dst.ptr + v0 * a.dst.C + c0);
                if (a.dst.accumulate) {
                    const uint4 r0 = q[0], r1 = q[1];
                    const unsigned w[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
                    for (int k = 0; k < 8; ++k) { g[2 * k] += __uint_as_float(w[k] << 16); g[2 * k + 1] += __uint_as_float(w[k] & 0xffff0000u); }
                }
                unsigned pk[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const bf16 lo = __float2bfloat16(g[2 * k]), hi = __float2bfloat16(g[2 * k + 1]);
                    pk[k] = (unsigned)__builtin_bit_cast(unsigned short, lo) | ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
                }
                q[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]); q[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
            } else {
                float* q = (float*)a.dst.ptr + v0 * a.dst.C + c0;
#pragma unroll
                for (int k = 0; k < 16; ++k) q[k] = a.dst.accumulate ? q[k] + g[k] : g[k];
            }
        }
    }
    if (a.slab) {
        __syncthreads();               // the filter in LDS is no longer needed
        const int Cin = a.Cin, Cout = a.Cout, O = Cin * Cout;
        small_wgrad_epilogue<NACC>(acc, nchunk, sm, a.slab + (int64_t)blockIdx.x * (O + Cout), [=](int ch, int i) {
            if (i < CO * 16) { int c = i / 16, k = i % 16; return c < Cout ? c * Cin + ch * 16 + k : -1; }
            int c = i - CO * 16;
            return (ch == 0 && c < Cout) ? O + c : -1;
        });
    }
}

static int ilog2_exact(int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; }
static bool wgrad_head_ok(const ConvGeom& g, int nsrc) {
    return nsrc == 1 && g.ks == 1 && g.stride == 1 && g.Cout <= 8 && g.Cin % 16 == 0 && ilog2_exact(g.Cin / 16) >= 0 && g.Cin <= 256;
}
static bool wgrad_first_ok(const ConvGeom& g, int nsrc) {
    return nsrc == 1 && g.ks == 3 && g.stride == 1 && g.Cin == 1 && g.Cout % 4 == 0 && ilog2_exact(g.Cout / 4) >= 0 && g.Cout <= 64 &&
           g.W % 4 == 0 && g.W <= 256;
}
static int wgrad_reg_blocks(int64_t items4) { int64_t nb = (items4 + 1023) / 1024; return nb < 1 ? 1 : (nb > 512 ? 512 : (int)nb); }
static bool wgrad_rows_ok(const ConvGeom& g) {
    int k3 = g.ks * g.ks * g.ks;
    int WI = (g.Wo - 1) * g.stride + g.ks;
    size_t lds = ((size_t)g.ks * g.ks * WI * g.Cin + (size_t)g.Wo * g.Cout) * 4;
    return (int64_t)k3 * g.Cin * g.Cout <= 1024 && lds <= 48 * 1024;
}

static int wgrad_small_blocks(const ConvGeom& g) { int rows = g.Do * g.Ho; return rows < 1024 ? rows : 1024; }
bool wgrad_small_supported(const ConvGeom& g, int nsrc) { return wgrad_head_ok(g, nsrc) || wgrad_first_ok(g, nsrc) || wgrad_rows_ok(g); }
size_t wgrad_small_scratch_bytes(const ConvGeom& g) {
    int k3 = g.ks * g.ks * g.ks;
    int64_t So = (int64_t)g.Do * g.Ho * g.Wo;
    size_t O = (size_t)k3 * g.Cin * g.Cout;
    size_t rows = (size_t)wgrad_small_blocks(g) * O * 4 + bias_grad_scratch_bytes(g.Cout, So) + 256;
    size_t regs = (size_t)512 * (O + g.Cout) * 4 + 256;
    return rows > regs ? rows : regs;
}
void launch_conv_wgrad_small(int dtype, const ConvGeom& g, const SrcDesc* src, int nsrc, const void* dy, float* dw, float* db,
                             void* scratch, hipStream_t s) {
    const int k3 = g.ks * g.ks * g.ks;
    const int64_t O = (int64_t)k3 * g.Cin * g.Cout;
    const bool head = wgrad_head_ok(g, nsrc), first = !head && wgrad_first_ok(g, nsrc);
    if (head || first) {
        WgradRegArgs a;
        a.g = g; a.src = src[0]; a.dy = dy; a.slab = (float*)scratch;
        const int64_t S = (int64_t)g.D * g.H * g.W;
        int nb;
        if (head) {
            a.lc = ilog2_exact(g.Cin / 16);
            nb = wgrad_reg_blocks((S << a.lc) * 4);
            const int CO = g.Cout <= 2 ? 2 : g.Cout <= 4 ? 4 : g.Cout <= 6 ? 6 : 8;
            const size_t lds = small_wgrad_lds_bytes(CO * 17, 1 << a.lc);
            switch (CO) {
                case 2: UNET_DISPATCH(dtype, (k_wgrad_head<T, 2><<<nb, 256, lds, s>>>(a))); break;
                case 4: UNET_DISPATCH(dtype, (k_wgrad_head<T, 4><<<nb, 256, lds, s>>>(a))); break;
                case 6: UNET_DISPATCH(dtype, (k_wgrad_head<T, 6><<<nb, 256, lds, s>>>(a))); break;
                default: UNET_DISPATCH(dtype, (k_wgrad_head<T, 8><<<nb, 256, lds, s>>>(a))); break;
            }
        } else {
            a.lc = ilog2_exact(g.Cout / 4);
            const int tiles = g.D * ((g.H + FIRST_TY - 1) / FIRST_TY);
            nb = tiles < 512 ? tiles : 512;
            const size_t l0 = (size_t)3 * (FIRST_TY + 2) * ((g.W + 5) & ~3) * 4, l1 = small_wgrad_lds_bytes(112, 1 << a.lc);
            const size_t lds = l0 > l1 ? l0 : l1;
            UNET_DISPATCH(dtype, (k_wgrad_first<T><<<nb, 256, lds, s>>>(a)));
        }
        slab_reduce2(a.slab, nb, O + g.Cout, dw, O, db, s);
        return;
    }
    WgradSmallArgs a;
    a.g = g; a.nsrc = nsrc; a.src[0] = src[0]; if (nsrc > 1) a.src[1] = src[1];
    a.dy = dy; a.slab = (float*)scratch;
    int nb = wgrad_small_blocks(g);
    int WI = (g.Wo - 1) * g.stride + g.ks;
    size_t lds = ((size_t)g.ks * g.ks * WI * g.Cin + (size_t)g.Wo * g.Cout) * 4;
    UNET_DISPATCH(dtype, (k_wgrad_small<T><<<nb, 256, lds, s>>>(a)));
    slab_reduce(a.slab, nb, O, dw, s);
    if (db) launch_bias_grad(dtype, dy, g.Cout, (int64_t)g.Do * g.Ho * g.Wo, db, (float*)scratch + (size_t)nb * O, s);
}

// ---- heads: public launchers ----
bool head_supported(const ConvGeom& g, int nsrc) { return wgrad_head_ok(g, nsrc); }
size_t head_bwd_scratch_bytes(const ConvGeom& g) { return (size_t)512 * ((size_t)g.Cin * g.Cout + g.Cout) * 4 + 256; }
template <typename T> static void head_launch(bool bwd, int CO, int nb, size_t lds, const HeadArgs& a, hipStream_t s) {
    switch (CO) {
        case 2: if (bwd) k_head_bwd<T, 2><<<nb, 256, lds, s>>>(a); else k_head_fwd<T, 2><<<nb, 256, lds, s>>>(a); break;
        case 4: if (bwd) k_head_bwd<T, 4><<<nb, 256, lds, s>>>(a); else k_head_fwd<T, 4><<<nb, 256, lds, s>>>(a); break;
        case 6: if (bwd) k_head_bwd<T, 6><<<nb, 256, lds, s>>>(a); else k_head_fwd<T, 6><<<nb, 256, lds, s>>>(a); break;
        default: if (bwd) k_head_bwd<T, 8><<<nb, 256, lds, s>>>(a); else k_head_fwd<T, 8><<<nb, 256, lds, s>>>(a); break;
    }
}
static int head_co(int Cout) { return Cout <= 2 ? 2 : Cout <= 4 ? 4 : Cout <= 6 ? 6 : 8; }
void launch_head_fwd(int dtype, const ConvGeom& g, const SrcDesc& src, const float* w, const float* bias, void* y, float* out_ncdhw,
                     hipStream_t s) {
    HeadArgs a = {};
    a.src = src; a.w = w; a.bias = bias; a.Cin = g.Cin; a.Cout = g.Cout; a.lc = ilog2_exact(g.Cin / 16);
    a.S = (int64_t)g.D * g.H * g.W; a.y = y; a.out = out_ncdhw;
    const int CO = head_co(g.Cout);
    int64_t items = ((a.S + 63) & ~(int64_t)63) << a.lc;
    int64_t nb = (items + 255) / 256;
    if (nb > 2048) nb = 2048;
    UNET_DISPATCH(dtype, (head_launch<T>(false, CO, (int)nb, (size_t)(CO + 2) * g.Cin * 4, a, s)));
}
// reduce_stream (optional): the slab sum that finishes dW / db runs there instead of on `s` -- it feeds nothing on the caller's
// chain; the CALLER orders reduce_stream after this launch (an event) and keeps `scratch` untouched until the reduce has run
void launch_head_bwd(int dtype, const ConvGeom& g, const SrcDesc& src, const float* dy_ncdhw, const void* dy_cl, const float* w,
                     DstGrad dst, float* dw, float* db, void* scratch, hipStream_t s, bool defer_reduce) {
    HeadArgs a = {};
    a.src = src; a.w = w; a.Cin = g.Cin; a.Cout = g.Cout; a.lc = ilog2_exact(g.Cin / 16);
    a.S = (int64_t)g.D * g.H * g.W; a.dy_planes = dy_ncdhw; a.dy_cl = dy_cl; a.dst = dst;
    a.slab = dw ? (float*)scratch : nullptr;
    const int CO = head_co(g.Cout);
    const int nb = wgrad_reg_blocks((a.S << a.lc) * 4);   // one (voxel, chunk) item per thread until 512 blocks are reached
    const size_t l0 = (size_t)(CO + 2) * g.Cin * 4, l1 = small_wgrad_lds_bytes(CO * 17, 1 << a.lc);
    UNET_DISPATCH(dtype, (head_launch<T>(true, CO, nb, l0 > l1 ? l0 : l1, a, s)));
    if (dw && !defer_reduce) slab_reduce2(a.slab, nb, (int64_t)g.Cin * g.Cout + g.Cout, dw, (int64_t)g.Cin * g.Cout, db, s);
}
// the deferred half of launch_head_bwd(..., defer_reduce = true): dw / db += the slab rows, on `s`
void launch_head_bwd_reduce(const ConvGeom& g, float* dw, float* db, const void* scratch, hipStream_t s) {
    const int lc = ilog2_exact(g.Cin / 16);
    const int nb = wgrad_reg_blocks((((int64_t)g.D * g.H * g.W) << lc) * 4);
    slab_reduce2((const float*)scratch, nb, (int64_t)g.Cin * g.Cout + g.Cout, dw, (int64_t)g.Cin * g.Cout, db, s);
}

}  // namespace unet
