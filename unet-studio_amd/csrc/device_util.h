// Device helpers shared by the kernel files (gfx950 only).
#pragma once
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>
#include <cstdint>

#include "kernels.h"

namespace unet {

typedef __hip_bfloat16 bf16;

template <typename T> __device__ __forceinline__ float ld(const T* p, int64_t i);
template <> __device__ __forceinline__ float ld<float>(const float* p, int64_t i) { return p[i]; }
template <> __device__ __forceinline__ float ld<bf16>(const bf16* p, int64_t i) { return __bfloat162float(p[i]); }
template <typename T> __device__ __forceinline__ void st(T* p, int64_t i, float v);
template <> __device__ __forceinline__ void st<float>(float* p, int64_t i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void st<bf16>(bf16* p, int64_t i, float v) { p[i] = __float2bfloat16(v); }

// activations of unet.cpp:91-98
__device__ __forceinline__ float act_f(float v, int act) {
    switch (act) {
        case 1: return v > 0.f ? v : 0.f;
        case 2: return v > 0.f ? v : 0.01f * v;
        case 3: return v > 0.f ? v : expm1f(v);
        default: return v;
    }
}
__device__ __forceinline__ float act_d(float v, int act) {
    switch (act) {
        case 1: return v > 0.f ? 1.f : 0.f;
        case 2: return v > 0.f ? 1.f : 0.01f;
        case 3: return v > 0.f ? 1.f : expf(v);
        default: return 1.f;
    }
}

template <typename T> __device__ __forceinline__ float view_ld(const SrcDesc& s, int64_t voxel, int c) {
    float v = ld<T>((const T*)s.ptr, voxel * s.C + c);
    if (s.scale) v = v * s.scale[c] + s.shift[c];
    return act_f(v, s.act);
}

#define UNET_DISPATCH(dtype, CALL)                 \
    do {                                           \
        if ((dtype) == 0) { typedef float T; CALL; } \
        else { typedef bf16 T; CALL; }             \
    } while (0)

static inline unsigned cdiv64(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

}  // namespace unet
