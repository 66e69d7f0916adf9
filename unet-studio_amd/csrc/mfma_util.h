// Shared device helpers of the MFMA kernel files (gfx950).
#pragma once
#include <atomic>
#include <cstdint>
#include <cstdlib>

#include "device_util.h"

namespace unet {

// Dynamic-LDS opt-in of a kernel (> 64 KB needs hipFuncSetAttribute), once PER DEVICE: the C++ drop-in keeps the reference's model
// of one process driving several devices from threads (train.cpp:592-600, other_models), and the attribute is per device.  One
// bit per device ordinal; a racing second call sets the same value again, which is harmless.
inline void set_max_lds_once(std::atomic<uint64_t>& done, const void* fn, int bytes) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return;
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    done.fetch_or(bit, std::memory_order_release);
}

// A "polite" weight-gradient launch (ConvGeom::polite) asks for more than half of a CU's LDS, so that a CU holds ONE of its 4-wave
// blocks: one wave per SIMD, >= 256 VGPRs per SIMD and ~77 KB of LDS left for the kernels of the caller's stream.  Measured (time line of
// a step, profiles/r10a_stretch.txt): at two blocks per CU a 240-VGPR weight-gradient kernel fills every SIMD's register file, and a
// 5-us norm kernel of the caller's stream that became ready beside it waited 64 us for the first block to leave.
inline int polite_lds(int lds, int polite) {
    constexpr int want = 83000;
    return (polite && want > lds) ? want : lds;
}
// UNET_NO_SLIDING_WINDOW=1 (read once per process): every convolution, conv_trans and weight-gradient shape runs on the halo-tile kernels
// k_mfma_conv_p / k_mfma_conv_small / k_mfma_wgrad instead of the sliding-window kernels whose vector-memory waits are counted by hand
// (k_mfma_conv_z, _z16, _z32, k_mfma_wgrad_z, _zd, k_s2_scatter, k_s2_gather, k_s2_wgrad) -- the documented fallback named by the build
// check (tools/check_asm_loads.py), and the way the halo-tile kernels stay under test (tests/test_gpu_parity.py runs the op cases with it).
inline bool sliding_window_off() {
    static const bool off = getenv("UNET_NO_SLIDING_WINDOW") != nullptr;
    return off;
}

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

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

// 8 bf16 channels of one voxel through the consumer-side transform act(x*scale+shift)
__device__ __forceinline__ uint4 transform8(uint4 v, bool xf, const float* sc, const float* sh, int act) {
    if (!(xf || act)) return v;
    unsigned wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float lo = bf_lo(wv[e]), hi = bf_hi(wv[e]);
        if (xf) { lo = fmaf(lo, sc[2 * e], sh[2 * e]); hi = fmaf(hi, sc[2 * e + 1], sh[2 * e + 1]); }
        lo = act_f(lo, act); hi = act_f(hi, act);
        wv[e] = pack_bf16x2(lo, hi);
    }
    return make_uint4(wv[0], wv[1], wv[2], wv[3]);
}

// arguments of the MFMA conv kernels (kernels_mfma_conv.hip, kernels_mfma_conv_z16.hip)
struct MfmaConvArgs {
    ConvGeom g;          // GEMM view: Cin = contraction channels, Cout = rows; D,H,W = volume read; Do,Ho,Wo = grid the tiles cover
    SrcDesc src[2];
    int nsrc;
    const void* w;       // packed filter
    const float* bias;   // nullptr: none (indexed by destination channel)
    void* out[2];        // channels-last bf16 destinations (split at outC[0] channels)
    int outC[2];
    int out_acc[2];
    int nout;
    float* stats;        // [nblk][Cout][2] or nullptr
    int tiles_x, tiles_y, tiles_z;
    int sc_C;            // SC: destination channels per tap (rows = 8 * sc_C)
    int oD, oH, oW;      // destination volume
    // dgrad only (k_mfma_conv_z16): norm-backward statistics of the destination tensor's view in the epilogue.  bn_u = the RAW tensor the
    // gradient belongs to ([voxel][bn_C] bf16), bn_stat = its norm's {mean, rstd, scale, shift} (4 x bn_C floats), bn_act its activation;
    // bn_partial[gridDim.x][bn_C][2] receives {sum g, sum g * xhat}, g = dL/d(view) * act'(u * scale + shift) (what k_norm_bwd_stats8 computes)
    const void* bn_u = nullptr;
    const float* bn_stat = nullptr;
    float* bn_partial = nullptr;
    int bn_act = 0, bn_C = 0;
};

// work split of the sliding-window kernels: (y, x) columns of the footprint x z segments of zlen output planes
struct ZWork { int nseg, zlen, cols_x, cols_y; };
// sliding window for a single 16-channel chunk (kernels_mfma_conv_z16.hip); returns 0 if the geometry does not qualify, else the
// number of statistics rows (gridDim.x)
int launch_conv_z16(const MfmaConvArgs& a, hipStream_t s);
bool conv_z16_applies(const MfmaConvArgs& a);               // the geometry test of launch_conv_z16 alone
int launch_conv_z32(const MfmaConvArgs& a, hipStream_t s);     // the same for a single 32-channel chunk
// kernels_mfma_wgrad_zd.hip: k_mfma_wgrad_z's 4-wave configurations with LDS-DMA staging (called by launch_mfma_wgrad_z with its own work split)
bool launch_wgrad_zd(const ConvGeom& g, const SrcDesc* src, int nsrc, const void* dy, float* slab, float* bias_slab, int wk, int pa, int pb, int cols_x,
                     int cols_y, int nseg, int zlen, int gx, int gy, hipStream_t s, int polite);

}  // namespace unet
