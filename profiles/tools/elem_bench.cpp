// Times the element-wise / reduction kernels of the train step one by one (HIP events, 20 launches back to back) against the bytes
// each must move: the per-kernel HBM rates quoted in DESIGN.md section 4.   Build: bash profiles/tools/build_elem_bench.sh
// Run (GPU box): unet-studio_amd/csrc/build/elem_bench [size=128] [C=16]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#include "../../unet-studio_amd/csrc/kernels.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static float time_us(const std::function<void()>& f, int iters = 20) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    CK(hipGetLastError());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3f / iters;
}
static void report(const char* name, float us, double bytes) { printf("%-34s %8.1f us   %7.1f MB   %5.2f TB/s\n", name, us, bytes / 1e6, bytes / us / 1e6); }

int main(int argc, char** argv) {
    using namespace unet;
    const int n = argc > 1 ? atoi(argv[1]) : 128, C = argc > 2 ? atoi(argv[2]) : 16, OC = 6;
    const int64_t S = (int64_t)n * n * n;
    void *u, *g, *a;
    float *stat, *coef, *partial, *logits, *dlogits, *level_out, *totals, *gamma, *dgamma;
    int64_t* target;
    CK(hipMalloc(&u, S * C * 2)); CK(hipMalloc(&g, S * C * 2)); CK(hipMalloc(&a, S * C * 2));
    CK(hipMalloc(&stat, 4 * C * 4)); CK(hipMalloc(&coef, 3 * C * 4)); CK(hipMalloc(&partial, 8 << 20));
    CK(hipMalloc(&gamma, C * 4)); CK(hipMalloc(&dgamma, 2 * C * 4));
    CK(hipMalloc(&logits, S * OC * 4)); CK(hipMalloc(&dlogits, S * OC * 4)); CK(hipMalloc(&target, S * 8));
    CK(hipMalloc(&level_out, 64 * 4)); CK(hipMalloc(&totals, 16 * 4));
    std::vector<uint16_t> hu(S * C);
    for (size_t i = 0; i < hu.size(); ++i) hu[i] = (uint16_t)(0x3f00 + (i * 2654435761u >> 20 & 0xff)) ^ (uint16_t)((i & 1) << 15);
    CK(hipMemcpy(u, hu.data(), S * C * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(g, hu.data(), S * C * 2, hipMemcpyHostToDevice));
    std::vector<float> hs(4 * C, 0.5f), hl(S * OC);
    for (size_t i = 0; i < hl.size(); ++i) hl[i] = (float)((i * 2654435761u >> 16) & 0xff) / 64.f - 2.f;
    std::vector<int64_t> ht(S);
    for (int64_t i = 0; i < S; ++i) ht[i] = (i * 2654435761u >> 13) % OC;
    CK(hipMemcpy(stat, hs.data(), 4 * C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(coef, hs.data(), 3 * C * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(gamma, hs.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemset(dgamma, 0, 2 * C * 4));
    CK(hipMemcpy(logits, hl.data(), S * OC * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(target, ht.data(), S * 8, hipMemcpyHostToDevice));
    CK(hipMemset(totals, 0, 64)); CK(hipMemset(level_out, 0, 256));
    const int act = 1;
    printf("# %d^3 voxels, %d channels bf16, %d classes\n", n, C, OC);
    const double T = (double)S * C * 2;
    report("hipMemcpyAsync d2d (r+w)", time_us([&] { CK(hipMemcpyAsync(a, u, S * C * 2, hipMemcpyDeviceToDevice, 0)); }), 2 * T);
    SrcDesc src; src.ptr = u; src.C = C; src.scale = stat + 2 * C; src.shift = stat + 3 * C; src.act = act;
    report("apply_view (activated copy)", time_us([&] { launch_apply_view(1, src, a, S, 0); }), 2 * T);
    report("stats_partial (fwd, unfused form)", time_us([&] { launch_stats_partial(1, u, C, S, partial, 0); }), T);
    report("norm_finalize", time_us([&] { launch_norm_finalize(partial, stats_blocks(S), C, S, gamma, gamma, 1e-5, stat, nullptr, nullptr, 0.1, 0); }), 0);
    CK(hipMemcpy(stat, hs.data(), 4 * C * 4, hipMemcpyHostToDevice));
    report("norm_bwd_partial (statistics)", time_us([&] { launch_norm_bwd_partial(1, g, u, C, S, stat, act, partial, 0); }), 2 * T);
    report("norm_bwd_finalize", time_us([&] { launch_norm_bwd_finalize(partial, stats_blocks(S), C, S, gamma, stat, coef, dgamma, dgamma + C, 0); }), 0);
    CK(hipMemcpy(coef, hs.data(), 3 * C * 4, hipMemcpyHostToDevice));
    report("norm_bwd_apply", time_us([&] { launch_norm_bwd_apply(1, g, u, C, S, stat, coef, act, 0); }), 3 * T);
    report("loss_partial", time_us([&] { launch_loss_partial(logits, target, OC, S, 0, partial, 0); }), (double)S * (OC * 4 + 8));
    report("loss_finalize", time_us([&] { launch_loss_finalize(partial, loss_blocks(S), OC, 1.f, 7, level_out, totals, 1, 0); }), 0);
    report("loss_grad", time_us([&] { launch_loss_grad(logits, target, OC, S, 0, level_out, 1.f, 7, dlogits, 0); }), (double)S * (OC * 8 + 8));
    return 0;
}
