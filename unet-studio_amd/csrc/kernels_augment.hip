// On-GPU sample augmentation (include/unet_augment.h): what visual_perception_augmentation_cuda does with ~25 launches,
// a device displacement field and an H2D/D2H round trip per sample (visual_perception_augmentation.cu:282-544), as five
// HBM-bound passes over volumes that are already resident:
//
//   k_aug_scale      (only with "downsample")  two trilinear resamplings per channel                       .cu:315-331
//   k_aug_photo      cropping, z-truncation, noise, ambient / diffuse / specular light, in place: 1R + 1W   .cu:333-380
//   k_aug_view       lens + foci + perspective + affine evaluated per output voxel in registers (the displacement
//                    field of .cu:412-432 is never stored), trilinear / majority gather, clamp at 0, per-channel max  .cu:434-446
//   k_aug_bg_max     maxima of the 5 rubber stamps per channel and of the Perlin texture (their normalisers)  .cu:471-476,498-509
//   k_aug_bg_blend   normalise, blend the stamps and the texture into the background voxels, per-channel max  .cu:477-512
//   k_aug_final      the last normalisation (or the zero-background mask) + write back over image / label    .cu:454-456,515-519,529-530
//
// Maxima are order-independent, so the results are deterministic: block maximum by shuffles, then one unsigned atomicMax per
// block on the float's bit pattern (all candidates are >= 0).  Everything is fp32 with FMA contraction OFF, so the numpy
// restatement in oracle/augment_ref.py follows the same rounding steps.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>

#include "../../include/unet_augment.h"
#include "kernels.h"

#pragma clang fp contract(off)

namespace unet {

namespace {

constexpr int AUG_T = 256;

struct AugGeom {
    int W, H, D, C;
    int64_t N;
    int gx, gy, gz;   // tile grid of the gather passes
};

// Thread -> voxel maps.  The gather passes (view, stamps) read 8 corners around a ROTATED position: with a block on a
// 16 x 4 x 4 brick its 256 footprints overlap in a compact source region that stays in the CU's L1, where a 256-long row would
// touch hundreds of cache lines with no reuse between corners (the blend pass at 2 x 256^3: 12.8 ms by rows, 6.6 ms by bricks).  Bricks are numbered so
// that each XCD (blocks are dealt round-robin to the 8 XCDs) works through one contiguous z-range of the volume and its
// private L2 keeps that range's halo.
constexpr int TX = 16, TY = 4, TZ = 4;
static_assert(TX * TY * TZ == AUG_T, "brick = block");

__device__ __forceinline__ bool brick_voxel(const AugGeom& g, int& x, int& y, int& z) {
    const unsigned nb = gridDim.x, per = (nb + 7) / 8;
    const unsigned b = (blockIdx.x & 7) * per + (blockIdx.x >> 3);   // uniform
    if (b >= nb) { x = y = z = 0; return false; }                     // only when nb is not a multiple of 8: ids >= nb idle
    const unsigned bx = b % g.gx, r = b / g.gx, by = r % g.gy, bz = r / g.gy;
    x = bx * TX + (threadIdx.x & (TX - 1));
    y = by * TY + ((threadIdx.x / TX) & (TY - 1));
    z = bz * TZ + threadIdx.x / (TX * TY);
    return x < g.W && y < g.H && z < g.D;
}

// element-wise passes that need coordinates: a block = 64 (x) x 4 (y) at one z, launched on a 3-D grid
__device__ __forceinline__ bool row_voxel(int W, int H, int& x, int& y, int& z) {
    x = blockIdx.x * 64 + (threadIdx.x & 63);
    y = blockIdx.y * 4 + (threadIdx.x >> 6);
    z = blockIdx.z;
    return x < W && y < H;
}

struct AugPhoto {   // host-derived constants of the element-wise stage, as the reference's host wrappers derive them
    int crop; float crop_pos[3], crop_radius, crop_value;
    int trunc_top, trunc_bottom;
    int noise; float noise_mag; unsigned noise_seed;
    int ambient; float ambient_value;
    int diffuse; float diffuse_f[3], diffuse_center[3];          // .cu:89-93
    int specular; float specular_pos[3], specular_freq, specular_mag, specular_b;   // .cu:110-113
};

struct AugView {
    UnetAugAffine view;
    int has_perspective; float perspective[3], center[3];        // .cu:192
    int has_lens; float lens_center[3], lens_magnitude;           // .cu:131-136
    int n_foci; float foci_pos[UNET_AUG_MAX_FOCI][3], foci_radius[UNET_AUG_MAX_FOCI], foci_r5[UNET_AUG_MAX_FOCI],
        foci_pi_r[UNET_AUG_MAX_FOCI];                             // .cu:157-158
    int is_label;
};

struct AugBg {
    int rubber; UnetAugAffine stamp[UNET_AUG_STAMPS]; float stamp_mag[UNET_AUG_MAX_CHANNELS][UNET_AUG_STAMPS];
    int perlin; unsigned char perm[512]; float perlin_zoom, perlin_mag;
};

// The three tables hold arrays that the kernels index with run-time channel / focus numbers; passed by value that
// indexing would make every thread keep a private copy, so one tiny launch parks them in device memory and the passes
// read them through a uniform pointer (scalar loads).
struct AugTables {
    AugPhoto p;
    AugView a;
    AugBg b;
};
static_assert(sizeof(AugTables) % 4 == 0 && sizeof(AugTables) <= 3584, "must fit the kernel-argument segment");

__global__ void k_aug_upload(AugTables t, AugTables* dst) {
    const unsigned* s = (const unsigned*)&t;
    unsigned* d = (unsigned*)dst;
    for (unsigned i = threadIdx.x; i < sizeof(AugTables) / 4; i += blockDim.x) d[i] = s[i];
}

// reduction cells (unsigned bit patterns of non-negative floats)
enum { CELL_VIEW = 0, CELL_BLEND = UNET_AUG_MAX_CHANNELS, CELL_STAMP = 2 * UNET_AUG_MAX_CHANNELS,
       CELL_PERLIN = CELL_STAMP + UNET_AUG_MAX_CHANNELS * UNET_AUG_STAMPS, CELL_COUNT = CELL_PERLIN + 1 };

// Block maximum -> row[blockIdx.x]; k_aug_cells then folds a row into its cell.  (A first version sent one atomicMax per
// block straight to the cell: 131 k device-scope atomics on one address made the blend pass 6.6 ms instead of 0.8)
__device__ __forceinline__ void block_max_to(float* __restrict__ row, float v) {
    __shared__ float red[AUG_T / 64];
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    __syncthreads();   // `red` may still be read by a previous call
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = red[0];
        for (int i = 1; i < AUG_T / 64; ++i) m = fmaxf(m, red[i]);
        row[blockIdx.x] = m;
    }
}

// cells[cell_of(row)] = max over the row's per-block maxima; row r maps to first_cell + r, except a tail row (the last one,
// when tail_cell >= 0) which maps to tail_cell.  n_blocks is a multiple of 8 and rows are 16-byte aligned.
constexpr int CELLS_T = 1024;
__global__ void __launch_bounds__(CELLS_T) k_aug_cells(const float* __restrict__ partial, unsigned n_blocks, int first_cell, int n_rows,
                                                        int tail_cell, unsigned* __restrict__ cells) {
    __shared__ float red[CELLS_T / 64];
    const int r = blockIdx.x;
    const float4* row = (const float4*)(partial + (size_t)r * n_blocks);
    float v = 0.f;
    for (unsigned i = threadIdx.x; i < n_blocks / 4; i += CELLS_T) {
        float4 q = row[i];
        v = fmaxf(fmaxf(v, fmaxf(q.x, q.y)), fmaxf(q.z, q.w));
    }
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = red[0];
        for (int i = 1; i < CELLS_T / 64; ++i) m = fmaxf(m, red[i]);
        cells[(tail_cell >= 0 && r == n_rows - 1) ? tail_cell : first_cell + r] = __float_as_uint(m);
    }
}

__device__ __forceinline__ float cell_value(const unsigned* cells, int i) { return __uint_as_float(cells[i]); }

// U(0,1] from a counter: murmur3 finaliser of (index, seed)
__device__ __forceinline__ float hash_u01(uint64_t index, unsigned seed) {
    unsigned h = (unsigned)index ^ ((unsigned)(index >> 32) * 0x9E3779B9u) ^ (seed * 0x85EBCA6Bu + 0x27D4EB2Fu);
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return (float)((h >> 8) + 1u) * (1.0f / 16777216.0f);
}

struct Tri {   // trilinear footprint: the 8 corner offsets (corner i: bit 0 = x, bit 1 = y, bit 2 = z; upper neighbours clamped), fractions
    unsigned o[8];
    float tx, ty, tz;
    bool ok;
};

// 32-bit offsets inside one volume (launch_augment checks D*H*W < 2^31): the loads become base + 32-bit-offset accesses and the
// address arithmetic stays off the quarter-rate 64-bit multiplier
__device__ __forceinline__ Tri locate(float x, float y, float z, int W, int H, int D) {
    Tri t;
    // NaN positions (a distortion focus's own centre voxel, .cu:151: 0/0) fail these comparisons, as out-of-volume ones do
    t.ok = (x >= 0.f) && (y >= 0.f) && (z >= 0.f) && (x <= (float)(W - 1)) && (y <= (float)(H - 1)) && (z <= (float)(D - 1));
    if (!t.ok) return t;
    float fx = floorf(x), fy = floorf(y), fz = floorf(z);
    t.tx = x - fx; t.ty = y - fy; t.tz = z - fz;
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1), z1 = min(z0 + 1, D - 1);
    const unsigned r00 = (unsigned)(z0 * H + y0) * (unsigned)W, r10 = (unsigned)(z0 * H + y1) * (unsigned)W,
                   r01 = (unsigned)(z1 * H + y0) * (unsigned)W, r11 = (unsigned)(z1 * H + y1) * (unsigned)W;
    t.o[0] = r00 + x0; t.o[1] = r00 + x1; t.o[2] = r10 + x0; t.o[3] = r10 + x1;
    t.o[4] = r01 + x0; t.o[5] = r01 + x1; t.o[6] = r11 + x0; t.o[7] = r11 + x1;
    return t;
}

__device__ __forceinline__ float lerp1(float t, float a, float b) { return a + t * (b - a); }

__device__ __forceinline__ float trilinear(const Tri& t, const float* __restrict__ vol) {
    float c00 = lerp1(t.tx, vol[t.o[0]], vol[t.o[1]]);
    float c10 = lerp1(t.tx, vol[t.o[2]], vol[t.o[3]]);
    float c01 = lerp1(t.tx, vol[t.o[4]], vol[t.o[5]]);
    float c11 = lerp1(t.tx, vol[t.o[6]], vol[t.o[7]]);
    return lerp1(t.tz, lerp1(t.ty, c00, c10), lerp1(t.ty, c01, c11));
}

// label resampling for class ids: the value holding the largest total trilinear weight among the 8 corners
// (first corner wins ties; corner order x fastest)
__device__ __forceinline__ float majority(const Tri& t, const float* __restrict__ vol) {
    float v[8], w[8];
    float wx[2] = {1.0f - t.tx, t.tx}, wy[2] = {1.0f - t.ty, t.ty}, wz[2] = {1.0f - t.tz, t.tz};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        v[i] = vol[t.o[i]];
        w[i] = wx[i & 1] * wy[(i >> 1) & 1] * wz[i >> 2];
    }
    float best = v[0], best_score = -1.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += (v[i] == v[j]) ? w[i] : 0.f;
        if (s > best_score) { best_score = s; best = v[j]; }
    }
    return best;
}

__device__ __forceinline__ void apply_affine(const UnetAugAffine& a, float& x, float& y, float& z) {
    float nx = a.sr[0] * x + a.sr[1] * y + a.sr[2] * z + a.shift[0];
    float ny = a.sr[3] * x + a.sr[4] * y + a.sr[5] * z + a.shift[1];
    float nz = a.sr[6] * x + a.sr[7] * y + a.sr[8] * z + a.shift[2];
    x = nx; y = ny; z = nz;
}

// ---------------------------------------------------------------------------------------------------------------
// dst (dw,dh,dd) <- trilinear resampling of src (sw,sh,sd): position = index * (source size / destination size)
__global__ void __launch_bounds__(AUG_T) k_aug_scale(const float* __restrict__ src, float* __restrict__ dst, int sw, int sh,
                                                      int sd, int dw, int dh, int dd, float rx, float ry, float rz) {
    int x, y, z;
    if (!row_voxel(dw, dh, x, y, z)) return;
    const int64_t i = ((int64_t)z * dh + y) * dw + x;
    float px = fminf((float)x * rx, (float)(sw - 1)), py = fminf((float)y * ry, (float)(sh - 1)),
          pz = fminf((float)z * rz, (float)(sd - 1));
    Tri t = locate(px, py, pz, sw, sh, sd);
    dst[i] = trilinear(t, src);
}

__global__ void __launch_bounds__(AUG_T) k_aug_photo(AugGeom g, const AugTables* __restrict__ T, float* __restrict__ image,
                                                      float* __restrict__ label) {
    const AugPhoto& p = T->p;
    int x, y, z;
    if (!row_voxel(g.W, g.H, x, y, z)) return;
    const int64_t i = ((int64_t)z * g.H + y) * g.W + x;
    float lab = label[i];
    bool cropped = false;
    if (p.crop && lab != 0.f) {   // .cu:9-22
        float dx = (float)x - p.crop_pos[0], dy = (float)y - p.crop_pos[1], dz = (float)z - p.crop_pos[2];
        float len = sqrtf(dx * dx + dy * dy + dz * dz);
        cropped = !(dx > p.crop_radius || dy > p.crop_radius || dz > p.crop_radius) && !(len > p.crop_radius);
    }
    bool cut = z < p.trunc_bottom || z >= g.D - p.trunc_top;   // .cu:31-59
    float light = 1.f, spec = 1.f;
    if (p.diffuse) {   // .cu:84
        float d = ((float)x - p.diffuse_center[0]) * p.diffuse_f[0] + ((float)y - p.diffuse_center[1]) * p.diffuse_f[1] +
                  ((float)z - p.diffuse_center[2]) * p.diffuse_f[2];
        light = fmaxf(0.f, 1.0f + d);
    }
    if (p.specular) {  // .cu:104
        float dx = (float)x - p.specular_pos[0], dy = (float)y - p.specular_pos[1], dz = (float)z - p.specular_pos[2];
        float len = sqrtf(dx * dx + dy * dy + dz * dz);
        spec = (cosf(len * p.specular_freq) + 1.0f) * p.specular_mag + p.specular_b;
    }
    for (int c = 0; c < g.C; ++c) {
        int64_t j = (int64_t)c * g.N + i;
        float v = image[j];
        // the loop over channels at .cu:339-340 clears the label while it crops the first channel, so later channels
        // find no label left inside the sphere: only channel 0 takes the cropping value
        if (cropped && c == 0) v = p.crop_value;
        if (cut) v = 0.f;
        if (p.noise) v += p.noise_mag * hash_u01((uint64_t)j, p.noise_seed);
        if (p.ambient) v += p.ambient_value;
        if (p.diffuse) v *= light;
        if (p.specular) v *= spec;
        image[j] = v;
    }
    if (cropped || cut) label[i] = 0.f;
}

__device__ __forceinline__ void view_position(const AugView& a, int x, int y, int z, float& px, float& py, float& pz) {
    px = (float)x; py = (float)y; pz = (float)z;
    if (a.has_lens) {
        // lens_distortion_kernel .cu:118-127, then create_distortion_at_kernel .cu:140-154 per focus, summed in that order
        float dx = px - a.lens_center[0], dy = py - a.lens_center[1], dz = pz - a.lens_center[2];
        float k = -a.lens_magnitude * (dx * dx + dy * dy + dz * dz);
        float ax = dx * k, ay = dy * k, az = dz * k;
        for (int f = 0; f < a.n_foci; ++f) {
            float ex = px - a.foci_pos[f][0], ey = py - a.foci_pos[f][1], ez = pz - a.foci_pos[f][2];
            float r = a.foci_radius[f];
            if (ex > r || ey > r || ez > r) continue;
            float len = sqrtf(ex * ex + ey * ey + ez * ez);
            if (len > r) continue;
            float s = -a.foci_r5[f] * sinf(len * a.foci_pi_r[f]) / len;   // 0/0 at the focus itself, as in the reference
            ax += ex * s; ay += ey * s; az += ez * s;
        }
        px += ax; py += ay; pz += az;
    }
    if (a.has_perspective) {   // .cu:182-183
        float q = a.perspective[0] * (px - a.center[0]) + a.perspective[1] * (py - a.center[1]) +
                  a.perspective[2] * (pz - a.center[2]) + 1.0f;
        px /= q; py /= q; pz /= q;
    }
    apply_affine(a.view, px, py, pz);
}

__global__ void __launch_bounds__(AUG_T) k_aug_view(AugGeom g, const AugTables* __restrict__ T, const float* __restrict__ image,
                                                     const float* __restrict__ label, float* __restrict__ out,
                                                     float* __restrict__ out_label, float* __restrict__ partial) {
    const AugView& a = T->a;
    int x, y, z;
    const bool live = brick_voxel(g, x, y, z);
    const int64_t i = ((int64_t)z * g.H + y) * g.W + x;
    Tri t;
    t.ok = false;
    if (live) {
        float px, py, pz;
        view_position(a, x, y, z, px, py, pz);
        t = locate(px, py, pz, g.W, g.H, g.D);
        float lab = 0.f;
        if (t.ok) lab = a.is_label ? majority(t, label) : trilinear(t, label);
        out_label[i] = lab;
    }
    for (int c = 0; c < g.C; ++c) {
        float v = 0.f;
        if (live && t.ok) {
            v = fmaxf(trilinear(t, image + (int64_t)c * g.N), 0.f);   // lower_threshold, .cu:451
        }
        if (live) out[(int64_t)c * g.N + i] = v;
        block_max_to(partial + (size_t)c * gridDim.x, v);
    }
}

__device__ __forceinline__ float perlin_grad(int hash, float x, float y, float z) {   // .cu:204-209
    int h = hash & 15;
    float u = h < 8 ? x : y;
    float v = h < 4 ? y : (h == 12 || h == 14 ? x : z);
    return ((h & 1) ? -u : u) + ((h & 2) ? -v : v);
}
__device__ __forceinline__ float perlin_fade(float t) { return t * t * t * (t * (t * 6.0f - 15.0f) + 10.0f); }

__device__ __forceinline__ float perlin_at(const unsigned char* p, float x, float y, float z) {   // .cu:211-247
    float flx = floorf(x), fly = floorf(y), flz = floorf(z);
    int xi = (int)flx & 255, yi = (int)fly & 255, zi = (int)flz & 255;
    float xf = x - flx, yf = y - fly, zf = z - flz;
    float u = perlin_fade(xf), v = perlin_fade(yf), w = perlin_fade(zf);
    int a = p[xi] + yi, b = p[xi + 1] + yi;
    int aaa = p[p[a] + zi], aba = p[p[a + 1] + zi], aab = p[p[a] + zi + 1], abb = p[p[a + 1] + zi + 1];
    int baa = p[p[b] + zi], bba = p[p[b + 1] + zi], bab = p[p[b] + zi + 1], bbb = p[p[b + 1] + zi + 1];
    float x1 = lerp1(u, perlin_grad(aaa, xf, yf, zf), perlin_grad(baa, xf - 1, yf, zf));
    float x2 = lerp1(u, perlin_grad(aba, xf, yf - 1, zf), perlin_grad(bba, xf - 1, yf - 1, zf));
    float y1 = lerp1(v, x1, x2);
    x1 = lerp1(u, perlin_grad(aab, xf, yf, zf - 1), perlin_grad(bab, xf - 1, yf, zf - 1));
    x2 = lerp1(u, perlin_grad(abb, xf, yf - 1, zf - 1), perlin_grad(bbb, xf - 1, yf - 1, zf - 1));
    float y2 = lerp1(v, x1, x2);
    return lerp1(w, y1, y2);
}

__device__ __forceinline__ float perlin_texture(const AugBg& b, const unsigned char* perm, int x, int y, int z) {   // .cu:498-509
    float acc = 0.f, pw = 1.0f;
    for (int o = 0; o < 4; ++o) {
        float scale = b.perlin_zoom * pw;
        acc += perlin_at(perm, (float)x * scale, (float)y * scale, (float)z * scale) * pw;
        pw *= 0.5f;
    }
    acc *= 2.0f;
    return acc - floorf(acc);
}

// tipl::masking(image,label), .cu:469: the pre-view channels keep their labelled part only (in place, after the view pass
// has read them), so that a stamp gathers one volume instead of two
__global__ void __launch_bounds__(AUG_T) k_aug_mask(int64_t n, int channels, float* __restrict__ image, const float* __restrict__ label) {
    int64_t i = (int64_t)blockIdx.x * AUG_T + threadIdx.x;
    if (i >= n) return;
    if (label[i] == 0.f)
        for (int c = 0; c < channels; ++c) image[(int64_t)c * n + i] = 0.f;
}

// a rubber stamp = a masked pre-view channel resampled through stamp s and clamped at 0 (.cu:472-475)
__device__ __forceinline__ Tri stamp_footprint(const AugGeom& g, const UnetAugAffine& s, int x, int y, int z) {
    float px = (float)x, py = (float)y, pz = (float)z;
    apply_affine(s, px, py, pz);
    return locate(px, py, pz, g.W, g.H, g.D);
}
__device__ __forceinline__ float stamp_value(const Tri& t, const float* __restrict__ im) {
    return t.ok ? fmaxf(trilinear(t, im), 0.f) : 0.f;
}

__global__ void __launch_bounds__(AUG_T) k_aug_bg_max(AugGeom g, const AugTables* __restrict__ T, const float* __restrict__ image,
                                                       float* __restrict__ partial) {
    const AugBg& b = T->b;
    __shared__ unsigned char perm[512];
    for (int k = threadIdx.x; k < 512; k += AUG_T) perm[k] = b.perm[k];
    __syncthreads();
    int x, y, z;
    const bool live = brick_voxel(g, x, y, z);
    if (b.rubber)
        for (int s = 0; s < UNET_AUG_STAMPS; ++s) {
            Tri t;
            t.ok = false;
            if (live) t = stamp_footprint(g, b.stamp[s], x, y, z);
            for (int c = 0; c < g.C; ++c)
                block_max_to(partial + (size_t)(c * UNET_AUG_STAMPS + s) * gridDim.x, stamp_value(t, image + (int64_t)c * g.N));
        }
    if (b.perlin) block_max_to(partial + (size_t)(g.C * UNET_AUG_STAMPS) * gridDim.x, live ? perlin_texture(b, perm, x, y, z) : 0.f);
}

__device__ __forceinline__ float normalised(float v, float mx, float upper) { return mx > 0.f ? v / mx * upper : v; }

__global__ void __launch_bounds__(AUG_T) k_aug_bg_blend(AugGeom g, const AugTables* __restrict__ T, const float* __restrict__ image,
                                                         float* __restrict__ out,
                                                         const float* __restrict__ out_label, const unsigned* __restrict__ cells,
                                                         float* __restrict__ partial) {
    const AugBg& b = T->b;
    __shared__ unsigned char perm[512];
    for (int k = threadIdx.x; k < 512; k += AUG_T) perm[k] = b.perm[k];
    __syncthreads();
    int x, y, z;
    const bool live = brick_voxel(g, x, y, z);
    const int64_t i = ((int64_t)z * g.H + y) * g.W + x;
    bool bgvox = false;
    float tex = 0.f;
    if (live) {
        bgvox = out_label[i] == 0.f;   // blend_kernel .cu:191-198
        if (bgvox && b.perlin) tex = normalised(perlin_texture(b, perm, x, y, z), cell_value(cells, CELL_PERLIN), b.perlin_mag);
    }

    for (int c = 0; c < g.C; ++c) {
        float v = 0.f;
        if (live) {
            int64_t j = (int64_t)c * g.N + i;
            v = normalised(out[j], cell_value(cells, CELL_VIEW + c), 1.0f);   // tipl::normalize after the view, .cu:451-452
            if (bgvox) {
                if (b.rubber)
                    for (int s = 0; s < UNET_AUG_STAMPS; ++s) {
                        float bg = stamp_value(stamp_footprint(g, b.stamp[s], x, y, z), image + (int64_t)c * g.N);
                        bg = normalised(bg, cell_value(cells, CELL_STAMP + c * UNET_AUG_STAMPS + s), b.stamp_mag[c][s]);
                        v += bg * fmaxf(0.1f, 1.0f - v);
                    }
                if (b.perlin) v += tex * fmaxf(0.1f, 1.0f - v);
            }
            v = fmaxf(v, 0.f);   // .cu:517
            out[j] = v;
        }
        block_max_to(partial + (size_t)c * gridDim.x, v);
    }
}

// mode 0: image = out / max(view)            (no background stage)
// mode 1: image = label ? out / max(view) : 0   (zero_background, .cu:452-457: no second normalisation)
// mode 2: image = out / max(blend)           (out already holds the blended, view-normalised values)
__global__ void __launch_bounds__(AUG_T) k_aug_final(AugGeom g, int mode, const float* __restrict__ out,
                                                      const float* __restrict__ out_label, float* __restrict__ image,
                                                      float* __restrict__ label, const unsigned* __restrict__ cells) {
    int64_t i = (int64_t)blockIdx.x * AUG_T + threadIdx.x;
    if (i >= g.N) return;
    float lab = out_label[i];
    label[i] = lab;
    for (int c = 0; c < g.C; ++c) {
        int64_t j = (int64_t)c * g.N + i;
        float v = normalised(out[j], cell_value(cells, (mode == 2 ? CELL_BLEND : CELL_VIEW) + c), 1.0f);
        if (mode == 1 && lab == 0.f) v = 0.f;
        image[j] = v;
    }
}

size_t aug_align(size_t v) { return (v + 255) / 256 * 256; }

unsigned brick_grid(const UnetAugmentRecipe& r) {   // rounded up to a multiple of 8, see brick_voxel
    int64_t b = (int64_t)((r.dims[0] + TX - 1) / TX) * ((r.dims[1] + TY - 1) / TY) * ((r.dims[2] + TZ - 1) / TZ);
    return (unsigned)((b + 7) / 8 * 8);
}
size_t partial_floats(const UnetAugmentRecipe& r) { return (size_t)(r.channels * UNET_AUG_STAMPS + 1) * brick_grid(r); }

}  // namespace

size_t augment_scratch_bytes(const UnetAugmentRecipe& r) {
    size_t n = (size_t)r.dims[0] * r.dims[1] * r.dims[2];
    size_t low = r.downsample ? (size_t)r.low_dims[0] * r.low_dims[1] * r.low_dims[2] : 0;
    return aug_align(CELL_COUNT * sizeof(unsigned)) + aug_align(sizeof(AugTables)) + aug_align(n * sizeof(float) * r.channels) +
           aug_align(n * sizeof(float)) + aug_align(low * sizeof(float)) + aug_align(partial_floats(r) * sizeof(float));
}

// Host side of the reference's *_cuda wrappers: the constants they derive before launching (double where the reference's
// expression is double, e.g. std::acos(-1)*0.5f/max, then rounded to the float kernel argument).
void launch_augment(const UnetAugmentRecipe& r, float* image, float* label, void* scratch, hipStream_t st) {
    AugGeom g{r.dims[0], r.dims[1], r.dims[2], r.channels, (int64_t)r.dims[0] * r.dims[1] * r.dims[2], 0, 0, 0};
    g.gx = (g.W + TX - 1) / TX; g.gy = (g.H + TY - 1) / TY; g.gz = (g.D + TZ - 1) / TZ;
    const int64_t bricks64 = (int64_t)g.gx * g.gy * g.gz;
    if (bricks64 > (int64_t)1 << 30 || g.N >= (int64_t)1 << 31) throw std::runtime_error("unet_augment: volume too large (D*H*W must stay below 2^31)");
    // ids are dealt to XCDs round-robin; rounding the grid up to a multiple of 8 keeps brick_voxel's renumbering onto
    const unsigned bricks = brick_grid(r);
    auto rows = [](int w, int h, int d) { return dim3((unsigned)((w + 63) / 64), (unsigned)((h + 3) / 4), (unsigned)d); };
    if (g.H > 4 * 65535 || g.D > 65535) throw std::runtime_error("unet_augment: height / depth beyond the launch grid");
    const int maxdim = std::max(r.dims[0], std::max(r.dims[1], r.dims[2]));
    char* base = (char*)scratch;
    unsigned* cells = (unsigned*)base;
    base += aug_align(CELL_COUNT * sizeof(unsigned));
    AugTables* tab = (AugTables*)base;
    base += aug_align(sizeof(AugTables));
    float* out = (float*)base;
    base += aug_align(g.N * sizeof(float) * r.channels);
    float* out_label = (float*)base;
    base += aug_align(g.N * sizeof(float));
    float* low = (float*)base;
    base += aug_align((r.downsample ? (size_t)r.low_dims[0] * r.low_dims[1] * r.low_dims[2] : 0) * sizeof(float));
    float* partial = (float*)base;
    if (hipError_t e = hipMemsetAsync(cells, 0, CELL_COUNT * sizeof(unsigned), st); e != hipSuccess)
        throw std::runtime_error(std::string("unet_augment: hipMemsetAsync: ") + hipGetErrorString(e));
    const unsigned nb = (unsigned)((g.N + AUG_T - 1) / AUG_T);

    if (r.downsample) {   // .cu:315-331: tipl::scale down and back up, per channel
        const int lw = r.low_dims[0], lh = r.low_dims[1], ld = r.low_dims[2];
        for (int c = 0; c < r.channels; ++c) {
            float* im = image + (int64_t)c * g.N;
            k_aug_scale<<<rows(lw, lh, ld), AUG_T, 0, st>>>(im, low, g.W, g.H, g.D, lw, lh, ld, (float)g.W / (float)lw,
                                                                              (float)g.H / (float)lh, (float)g.D / (float)ld);
            k_aug_scale<<<rows(g.W, g.H, g.D), AUG_T, 0, st>>>(low, im, lw, lh, ld, g.W, g.H, g.D, (float)lw / (float)g.W, (float)lh / (float)g.H,
                                              (float)ld / (float)g.D);
        }
    }

    AugTables tables{};
    AugPhoto& p = tables.p;
    p.crop = r.crop; p.crop_radius = r.crop_radius; p.crop_value = r.crop_value;
    for (int k = 0; k < 3; ++k) p.crop_pos[k] = (float)r.crop_pos[k];
    p.trunc_top = r.trunc_top; p.trunc_bottom = r.trunc_bottom;
    p.noise = r.noise; p.noise_mag = r.noise_mag; p.noise_seed = r.noise_seed;
    p.ambient = r.ambient; p.ambient_value = r.ambient_value;
    p.diffuse = r.diffuse;
    if (r.diffuse) {   // diffuse_light_cuda .cu:89-93
        float fx = r.diffuse_dir[0], fy = r.diffuse_dir[1], fz = r.diffuse_dir[2];
        float len = std::sqrt(fx * fx + fy * fy + fz * fz);
        if (len != 0.f) { fx /= len; fy /= len; fz /= len; }
        float k = r.diffuse_mag / (float)maxdim;
        p.diffuse_f[0] = fx * k; p.diffuse_f[1] = fy * k; p.diffuse_f[2] = fz * k;
        for (int d = 0; d < 3; ++d) p.diffuse_center[d] = (float)r.dims[d] * 0.5f;
    }
    p.specular = r.specular;
    if (r.specular) {  // specular_light_cuda .cu:110-113
        for (int k = 0; k < 3; ++k) p.specular_pos[k] = (float)r.specular_pos[k];
        p.specular_mag = r.specular_mag;
        p.specular_b = 1.0f - r.specular_mag - r.specular_mag;
        p.specular_freq = (float)((double)r.specular_freq * (std::acos(-1.0) * 0.5 / (double)maxdim));
    }
    AugView& a = tables.a;
    a.view = r.view;
    a.has_perspective = r.has_perspective;
    a.has_lens = r.has_lens;
    a.is_label = r.is_label;
    for (int d = 0; d < 3; ++d) {
        a.perspective[d] = r.perspective[d];
        a.center[d] = (float)r.dims[d] / 2.0f;            // .cu:192
        a.lens_center[d] = (float)(r.dims[d] / 2);        // integer halves, .cu:133-134
    }
    {   // lens_distortion_cuda .cu:129-138: radius = max/2 in integers
        float radius = (float)(maxdim / 2);
        a.lens_magnitude = r.lens_magnitude / (radius * radius);
    }
    a.n_foci = r.has_lens ? r.n_foci : 0;
    for (int f = 0; f < a.n_foci; ++f) {   // create_distortion_at_cuda .cu:155-161
        for (int d = 0; d < 3; ++d) a.foci_pos[f][d] = (float)r.foci_pos[f][d];
        a.foci_radius[f] = r.foci_radius[f];
        a.foci_r5[f] = r.foci_radius[f] * r.foci_magnitude[f];
        a.foci_pi_r[f] = (float)(std::acos(-1.0) / (double)r.foci_radius[f]);
    }
    // background stage: with a label and without zero_background the reference normalises a second time (.cu:515-519); when
    // nothing was blended that divides by exactly 1 (the view normalisation left max == 1), so only a real blend needs mode 2
    int mode = 0;
    AugBg& b = tables.b;
    if (r.is_label && r.zero_background) mode = 1;
    else if (r.is_label && (r.rubber || r.perlin)) {
        mode = 2;
        b.rubber = r.rubber; b.perlin = r.perlin;
        for (int s = 0; s < UNET_AUG_STAMPS; ++s) b.stamp[s] = r.stamp[s];
        for (int c = 0; c < UNET_AUG_MAX_CHANNELS; ++c)
            for (int s = 0; s < UNET_AUG_STAMPS; ++s) b.stamp_mag[c][s] = r.stamp_mag[c][s];
        for (int k = 0; k < 512; ++k) b.perm[k] = r.perm[k];
        b.perlin_zoom = r.perlin_zoom; b.perlin_mag = r.perlin_mag;
    }

    k_aug_upload<<<1, 64, 0, st>>>(tables, tab);
    if (r.crop || r.trunc_top || r.trunc_bottom || r.noise || r.ambient || r.diffuse || r.specular)
        k_aug_photo<<<rows(g.W, g.H, g.D), AUG_T, 0, st>>>(g, tab, image, label);
    k_aug_view<<<bricks, AUG_T, 0, st>>>(g, tab, image, label, out, out_label, partial);
    k_aug_cells<<<g.C, CELLS_T, 0, st>>>(partial, bricks, CELL_VIEW, g.C, -1, cells);
    if (mode == 2) {
        const int rows_bg = g.C * UNET_AUG_STAMPS + 1;
        if (r.rubber) k_aug_mask<<<nb, AUG_T, 0, st>>>(g.N, g.C, image, label);
        if (!(r.rubber && r.perlin))   // rows the pass does not write must read as 0
            if (hipError_t e = hipMemsetAsync(partial, 0, (size_t)rows_bg * bricks * sizeof(float), st); e != hipSuccess)
                throw std::runtime_error(std::string("unet_augment: hipMemsetAsync: ") + hipGetErrorString(e));
        k_aug_bg_max<<<bricks, AUG_T, 0, st>>>(g, tab, image, partial);
        k_aug_cells<<<rows_bg, CELLS_T, 0, st>>>(partial, bricks, CELL_STAMP, rows_bg, CELL_PERLIN, cells);
        k_aug_bg_blend<<<bricks, AUG_T, 0, st>>>(g, tab, image, out, out_label, cells, partial);
        k_aug_cells<<<g.C, CELLS_T, 0, st>>>(partial, bricks, CELL_BLEND, g.C, -1, cells);
    }
    k_aug_final<<<nb, AUG_T, 0, st>>>(g, mode, out, out_label, image, label, cells);
}

// ================================================================================================================
// simulate_modality (include/unet_augment.h; train.cpp:43-178): three passes over a resident volume
//   k_sim_smooth x2   tipl::filter::gaussian twice (train.cpp:62-63): here the 3x3x3 binomial (1,2,1)^3/64 with the border voxels
//                     replicated -- TIPL's kernel is not in the reference tree (parity unpinned); the first pass reads the tissue
//                     volume through the label look-up table (train.cpp:59-60) instead of materialising it
//   k_sim_remap       the 20-term polynomial + pow (train.cpp:84-104), per-block min / max over the qualifying voxels
//   k_sim_final       the stretch to [0,1] and clamp (train.cpp:110-115)
// ================================================================================================================
namespace {

struct SimTables {
    float lut[UNET_SIM_MAX_LABELS];
    float term_w[UNET_SIM_TERMS];
    unsigned char ta[UNET_SIM_TERMS], tb[UNET_SIM_TERMS], tc[UNET_SIM_TERMS], td[UNET_SIM_TERMS];
    float gamma;
    int with_label, max_label;
};

__global__ void k_sim_upload(SimTables t, SimTables* dst) {
    const unsigned* s = (const unsigned*)&t;
    unsigned* d = (unsigned*)dst;
    for (unsigned i = threadIdx.x; i < sizeof(SimTables) / 4; i += blockDim.x) d[i] = s[i];
}
static_assert(sizeof(SimTables) % 4 == 0, "word copy");

// dst = binomial 3x3x3 of src'; src' = lut[label] (FIRST && with_label), else src
template <bool FIRST>
__global__ void __launch_bounds__(AUG_T) k_sim_smooth(int W, int H, int D, const SimTables* __restrict__ T, const float* __restrict__ src,
                                                       float* __restrict__ dst) {
    __shared__ float lds_lut[UNET_SIM_MAX_LABELS];
    const bool lut = FIRST && T->with_label;
    if (lut) {   // 27 look-ups per voxel with data-dependent indices: from LDS, not from the table in global memory
        for (int k = threadIdx.x; k < UNET_SIM_MAX_LABELS; k += AUG_T) lds_lut[k] = T->lut[k];
        __syncthreads();
    }
    int x, y, z;
    if (!row_voxel(W, H, x, y, z)) return;
    float acc = 0.f;
#pragma unroll
    for (int kz = 0; kz < 3; ++kz) {
        const int zz = min(max(z + kz - 1, 0), D - 1);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = min(max(y + ky - 1, 0), H - 1);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int xx = min(max(x + kx - 1, 0), W - 1);
                float v = src[(unsigned)((zz * H + yy) * W + xx)];     // volumes < 2^31 voxels (checked by the launcher)
                if (lut) v = lds_lut[min(max((int)v, 0), T->max_label)];
                const float w = (float)((kz == 1 ? 2 : 1) * (ky == 1 ? 2 : 1) * (kx == 1 ? 2 : 1)) * (1.0f / 64.0f);
                acc += w * v;
            }
        }
    }
    dst[((int64_t)z * H + y) * W + x] = acc;
}

__global__ void __launch_bounds__(AUG_T) k_sim_remap(int64_t n, const SimTables* __restrict__ T, float* __restrict__ t1w,
                                                      const float* __restrict__ tissue, const float* __restrict__ label,
                                                      float* __restrict__ partial) {
    __shared__ float red[2][AUG_T / 64];
    const int64_t i = (int64_t)blockIdx.x * AUG_T + threadIdx.x;
    float mn = 3.402823466e+38f, mx = -3.402823466e+38f;
    if (i < n) {
        const float x = t1w[i];
        float out = 0.f;
        if (!(x <= 0.02f)) {
            const float z = tissue[i], rx = 1.0f - x, rz = 1.0f - z;
            const float px[4] = {1.0f, x, x * x, x * x * x}, pz[4] = {1.0f, z, z * z, z * z * z};
            const float qx[4] = {1.0f, rx, rx * rx, rx * rx * rx}, qz[4] = {1.0f, rz, rz * rz, rz * rz * rz};
            float s = 0.f;
            for (int k = 0; k < UNET_SIM_TERMS; ++k) {
                // selects instead of indexing the small arrays with run-time exponents (that would put them in scratch)
                const int a = T->ta[k], b = T->tb[k], c = T->tc[k], d = T->td[k];
                const float fa = a == 0 ? px[0] : a == 1 ? px[1] : a == 2 ? px[2] : px[3];
                const float fb = b == 0 ? pz[0] : b == 1 ? pz[1] : b == 2 ? pz[2] : pz[3];
                const float fc = c == 0 ? qx[0] : c == 1 ? qx[1] : c == 2 ? qx[2] : qx[3];
                const float fd = d == 0 ? qz[0] : d == 1 ? qz[1] : d == 2 ? qz[2] : qz[3];
                s += T->term_w[k] * fa * fb * fc * fd;
            }
            out = powf(s, T->gamma);
            if (!T->with_label || label[i] != 0.f) { mn = out; mx = out; }     // train.cpp:105-109 / :168-169
        }
        t1w[i] = out;
    }
    for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = mn; red[1][threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < AUG_T / 64; ++k) { mn = fminf(mn, red[0][k]); mx = fmaxf(mx, red[1][k]); }
        mn = fminf(mn, red[0][0]); mx = fmaxf(mx, red[1][0]);
        partial[2 * (size_t)blockIdx.x] = mn;
        partial[2 * (size_t)blockIdx.x + 1] = mx;
    }
}

__global__ void __launch_bounds__(CELLS_T) k_sim_minmax(const float* __restrict__ partial, unsigned n_blocks, float* __restrict__ mm) {
    __shared__ float red[2][CELLS_T / 64];
    float mn = 3.402823466e+38f, mx = -3.402823466e+38f;
    for (unsigned i = threadIdx.x; i < n_blocks; i += CELLS_T) { mn = fminf(mn, partial[2 * (size_t)i]); mx = fmaxf(mx, partial[2 * (size_t)i + 1]); }
    for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = mn; red[1][threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 0; k < CELLS_T / 64; ++k) { mn = fminf(mn, red[0][k]); mx = fmaxf(mx, red[1][k]); }
        mm[0] = mn; mm[1] = mx;
    }
}

__global__ void __launch_bounds__(AUG_T) k_sim_final(int64_t n, const float* __restrict__ mm, float* __restrict__ t1w) {
    const int64_t i = (int64_t)blockIdx.x * AUG_T + threadIdx.x;
    if (i >= n) return;
    const float mn = mm[0], mx = mm[1];
    if (!(mx > mn)) return;
    const float inv = 1.0f / (mx - mn);
    float v = (t1w[i] - mn) * inv;                       // t1w -= mn; t1w *= 1/(mx-mn), train.cpp:112-113
    t1w[i] = fminf(fmaxf(v, 0.f), 1.0f);                 // tipl::upper_lower_threshold(t1w, 0, 1)
}

}  // namespace

size_t simulate_scratch_bytes(const UnetSimulateRecipe& r) {
    const size_t n = (size_t)r.dims[0] * r.dims[1] * r.dims[2];
    const size_t nb = (n + AUG_T - 1) / AUG_T;
    return aug_align(sizeof(SimTables)) + 2 * aug_align(n * sizeof(float)) + aug_align(2 * nb * sizeof(float)) + 256;
}

void launch_simulate_modality(const UnetSimulateRecipe& r, float* t1w, const float* label, void* scratch, hipStream_t st) {
    const int W = r.dims[0], H = r.dims[1], D = r.dims[2];
    const int64_t n = (int64_t)W * H * D;
    const unsigned nb = (unsigned)((n + AUG_T - 1) / AUG_T);
    char* base = (char*)scratch;
    SimTables* tab = (SimTables*)base; base += aug_align(sizeof(SimTables));
    float* ta = (float*)base; base += aug_align(n * sizeof(float));
    float* tb = (float*)base; base += aug_align(n * sizeof(float));
    float* partial = (float*)base; base += aug_align(2 * (size_t)nb * sizeof(float));
    float* mm = (float*)base;
    SimTables t{};
    for (int k = 0; k < UNET_SIM_MAX_LABELS; ++k) t.lut[k] = r.lut[k];
    for (int k = 0; k < UNET_SIM_TERMS; ++k) {
        t.term_w[k] = r.term_w[k]; t.ta[k] = r.term_a[k]; t.tb[k] = r.term_b[k]; t.tc[k] = r.term_c[k]; t.td[k] = r.term_d[k];
    }
    t.gamma = r.gamma; t.with_label = r.with_label; t.max_label = r.max_label;
    k_sim_upload<<<1, 64, 0, st>>>(t, tab);
    if (H > 4 * 65535 || D > 65535 || n >= ((int64_t)1 << 31)) throw std::runtime_error("unet_simulate_modality: volume beyond the launch grid / 2^31 voxels");
    const dim3 rows((unsigned)((W + 63) / 64), (unsigned)((H + 3) / 4), (unsigned)D);
    k_sim_smooth<true><<<rows, AUG_T, 0, st>>>(W, H, D, tab, r.with_label ? label : t1w, ta);
    k_sim_smooth<false><<<rows, AUG_T, 0, st>>>(W, H, D, tab, ta, tb);
    k_sim_remap<<<nb, AUG_T, 0, st>>>(n, tab, t1w, tb, label, partial);
    k_sim_minmax<<<1, CELLS_T, 0, st>>>(partial, nb, mm);
    k_sim_final<<<nb, AUG_T, 0, st>>>(n, mm, t1w);
}

}  // namespace unet
