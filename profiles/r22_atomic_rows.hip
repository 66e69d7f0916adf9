// micro-benchmark: what would it cost to fold the norm statistics' block partials with 64-bit integer atomics (deterministic: integer
// addition commutes) into R rows instead of one row per block + a finalize launch?  512 blocks x 256 threads, each block adds 32 (or
// 64) values to row blockIdx % R at the end of ~20 us of streaming work.   hipcc --offload-arch=gfx950 -O3 r22_atomic_rows.hip -o t && ./t
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float4* __restrict__ in, float4* __restrict__ out, long long* acc, int R, int nval, int iters, int mode) {
    float4 s = make_float4(0, 0, 0, 0);
    const size_t base = (size_t)blockIdx.x * iters * 256 + threadIdx.x;
    for (int i = 0; i < iters; ++i) { float4 v = in[base + (size_t)i * 256]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; out[base + (size_t)i * 256] = v; }
    __shared__ float red[256];
    red[threadIdx.x] = s.x + s.y + s.z + s.w;
    __syncthreads();
    if (mode == 1 && threadIdx.x < nval) {
        long long q = (long long)(red[threadIdx.x] * 1048576.0f);
        atomicAdd((unsigned long long*)&acc[(size_t)(blockIdx.x % R) * nval + threadIdx.x], (unsigned long long)q);
    } else if (mode == 0 && threadIdx.x < nval) {
        ((float*)acc)[(size_t)blockIdx.x * nval + threadIdx.x] = red[threadIdx.x];
    }
}
int main() {
    const int nb = 512, iters = 64;   // 512 x 64 x 256 x 16 B = 134 MB read + 134 MB written
    float4 *in, *out; long long* acc;
    hipMalloc(&in, (size_t)nb * iters * 256 * 16); hipMalloc(&out, (size_t)nb * iters * 256 * 16); hipMalloc(&acc, 1 << 22);
    hipMemset(in, 0, (size_t)nb * iters * 256 * 16); hipMemset(acc, 0, 1 << 22);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](int R, int nval, int mode) {
        for (int w = 0; w < 3; ++w) k<<<nb, 256>>>(in, out, acc, R, nval, iters, mode);
        hipEventRecord(e0);
        for (int r = 0; r < 20; ++r) k<<<nb, 256>>>(in, out, acc, R, nval, iters, mode);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("mode %s  R %3d  values %2d : %.2f us per launch\n", mode ? "atomic" : "rows  ", R, nval, ms * 50.0f);
    };
    run(1, 32, 0); run(1, 64, 0);
    for (int nval : {32, 64}) for (int R : {1, 4, 8, 32, 128}) run(R, nval, 1);
    return 0;
}
