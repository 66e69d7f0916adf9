// Times the optimizer step of train_unet's thread C (train.cpp:604-766: forward, calc_losses over the deep-supervision levels, backward,
// /batch_size, clip_grad_norm_(12), SGD-Nesterov, zero_grad) through the C++ drop-in host -- include/unet.hpp + unet_host.cpp over
// libunet_hip.so -- on one synthetic sample resident in HBM: the number bench.py reports beside its own (Python-hosted) ms_per_step.
//   bench_host [size = 128] [steps = 50] [warmup = 10] [bf16|fp32]        prints one JSON line
#include <chrono>
#include <iostream>
#include <sstream>
#include <string>

#include <torch/torch.h>
#include <c10/hip/HIPFunctions.h>

#include "unet.hpp"

static std::string default_feature(int out_count) {      // train.cpp:1054-1069 (a data string: the reference's default architecture)
    const int ch[6] = {16, 32, 64, 128, 256, 256};
    const std::string nl = "norm,leaky_relu", out = "conv" + std::to_string(out_count) + ",ks1,stride1";
    std::ostringstream s;
    for (int i = 0; i < 6; ++i) {
        s << "conv" << ch[i] << ",ks3,stride" << (i == 0 ? 1 : 2) << "+" << nl << "+conv" << ch[i] << ",ks3,stride1+" << nl;
        if (i == 5) s << "+conv_trans256,ks2,stride2";
        s << "\n";
    }
    const int dec[4][2] = {{256, 128}, {128, 64}, {64, 32}, {32, 16}};
    for (auto& d : dec)
        s << "conv" << d[0] << ",ks3,stride1+" << nl << "+conv" << d[0] << ",ks3,stride1+" << nl << "+" << out << "+conv_trans" << d[1] << ",ks2,stride2\n";
    s << "conv16,ks3,stride1+" << nl << "+conv16,ks3,stride1+" << nl << "+" << out;
    return s.str();
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : 128, steps = argc > 2 ? std::atoi(argv[2]) : 50, warmup = argc > 3 ? std::atoi(argv[3]) : 10;
    const bool bf16 = !(argc > 4 && std::string(argv[4]) == "fp32");
    if (!torch::cuda::is_available()) { std::cerr << "bench_host needs a GPU" << std::endl; return 2; }
    try {
        torch::manual_seed(0);
        UNet3d model(1, 6, default_feature(6));
        model->engine_dtype = bf16 ? 1 : 0;
        torch::Device dev(torch::kCUDA, 0);
        model->to(dev);
        model->train();
        model->create_optimizer(0.001f);
        auto x = torch::rand({1, 1, n, n, n}).to(dev);
        auto t = torch::randint(0, 6, {1, n, n, n}, torch::kLong).to(dev);
        auto step = [&]() {
            auto losses = model->loss_and_backward(x, t, true, true, true);     // train.cpp:628-706 for one sample
            model->sgd_step(0.001f, 1.0f);                                      // train.cpp:759-766 (batch_size 1)
            return losses;
        };
        torch::Tensor last;
        for (int i = 0; i < warmup; ++i) last = step();
        c10::hip::device_synchronize();
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < steps; ++i) last = step();
        c10::hip::device_synchronize();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::cout << "{\"host\": \"C++ (include/unet.hpp + unet_host.cpp: loss_and_backward + sgd_step)\", \"size\": " << n << ", \"dtype\": \""
                  << (bf16 ? "bf16" : "fp32") << "\", \"steps\": " << steps << ", \"warmup\": " << warmup << ", \"ms_per_step\": " << dt / steps * 1e3
                  << ", \"value\": " << (double)n * n * n * steps / dt << ", \"unit\": \"voxels/s\", \"last_loss\": " << last[0].item<float>() << "}" << std::endl;
    } catch (const std::exception& e) {
        std::cerr << "bench_host: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
