// Does a CU-masked stream (hipExtStreamCreateWithCUMask) confine a launch to a subset of the CUs on this stack?
//   hipcc --offload-arch=gfx950 -O2 -o cu_mask_probe cu_mask_probe.hip && ./cu_mask_probe
// Every block records where it ran (XCC_ID and HW_ID: SE / CU) and a busy kernel is timed: with a working mask the number of
// distinct (xcc, se, cu) triples drops to the mask's population and the busy kernel slows down accordingly.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>

__global__ void where(uint32_t* out) {
    if (threadIdx.x == 0) {
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc;
    }
    // stay resident long enough that the launch spreads over everything it may use
    float x = threadIdx.x;
    for (int i = 0; i < 20000; ++i) x = x * 1.0001f + 0.5f;
    if (x == 12345.f) out[0] = 0;
}
__global__ void busy(float* out, int iters) {
    float x = threadIdx.x, y = blockIdx.x;
    for (int i = 0; i < iters; ++i) { x = x * 1.0001f + y; y = y * 0.9999f + x; }
    if (x == 12345.f) out[0] = y;
}

static void run(const char* name, hipStream_t s) {
    const int nb = 4096;
    uint32_t* d; hipMalloc(&d, nb * 8);
    where<<<nb, 256, 0, s>>>(d);
    hipStreamSynchronize(s);
    std::vector<uint32_t> h(nb * 2);
    hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
    std::set<uint32_t> cus; int per_xcc[8] = {0};
    for (int b = 0; b < nb; ++b) {
        uint32_t hw = h[b * 2], xcc = h[b * 2 + 1] & 0xf;
        uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        cus.insert((xcc << 16) | (se << 8) | (sh << 4) | cu);
        per_xcc[xcc & 7]++;
    }
    float* o; hipMalloc(&o, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    busy<<<8192, 256, 0, s>>>(o, 1000);
    hipEventRecord(e0, s);
    busy<<<8192, 256, 0, s>>>(o, 20000);
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s distinct (xcc,se,sh,cu): %3zu   blocks per xcc: %d %d %d %d %d %d %d %d   busy kernel %.3f ms\n", name, cus.size(),
           per_xcc[0], per_xcc[1], per_xcc[2], per_xcc[3], per_xcc[4], per_xcc[5], per_xcc[6], per_xcc[7], ms);
    hipFree(d); hipFree(o);
}

int main() {
    hipStream_t s0; hipStreamCreate(&s0);
    run("unmasked", s0);
    struct { const char* name; uint32_t w[10]; } masks[] = {
        {"first 128 bits", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0, 0, 0}},
        {"last 128 bits of 256", {0, 0, 0, 0, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0}},
        {"every other bit", {0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0, 0}},
        {"0x0f0f0f0f", {0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0, 0}},
        {"0x00ff00ff", {0x00ff00ffu, 0x00ff00ffu, 0x00ff00ffu, 0x00ff00ffu, 0x00ff00ffu, 0x00ff00ffu, 0x00ff00ffu, 0x00ff00ffu, 0, 0}},
        {"first 32 bits", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0, 0, 0}},
        {"first 64 bits", {0xffffffffu, 0xffffffffu, 0, 0, 0, 0, 0, 0, 0, 0}},
    };
    for (auto& m : masks) {
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, m.w);
        if (e != hipSuccess) { printf("%-28s hipExtStreamCreateWithCUMask: %s\n", m.name, hipGetErrorString(e)); continue; }
        run(m.name, s);
        hipStreamDestroy(s);
    }
    // environment route: HSA_CU_MASK / ROC_GLOBAL_CU_MASK are process-wide, not per stream: not tried here
    return 0;
}
