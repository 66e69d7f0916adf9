// GPU parity test of the C++ drop-in (include/unet.hpp + unet_host.cpp) against a libtorch CPU network assembled
// with the same torch::nn modules, in the same order, as the reference's unet.cpp:24-193 builds for this DSL string.
// Exit code 0 and "OK" on success.  Usage: test_unet_hpp [fp32|bf16]
#include "unet.hpp"
#include "unet_hip.h"
#include <c10/hip/HIPCachingAllocator.h>
#include <c10/hip/HIPStream.h>
#include <unistd.h>
#include <cstdio>
#include <atomic>
#include <iostream>
#include <thread>

static double rel(const torch::Tensor& a, const torch::Tensor& b) {
    auto A = a.to(torch::kCPU).to(torch::kFloat64), B = b.to(torch::kCPU).to(torch::kFloat64);
    return ((A - B).abs().max() / B.abs().max().clamp_min(1e-30)).item<double>();
}
#define REQUIRE(cond, msg) do { if (!(cond)) { std::cerr << "FAILED: " << msg << std::endl; return 1; } } while (0)

// reference-order network for:  conv8,ks3,stride1+norm,leaky_relu \n conv16,ks3,stride2+norm,leaky_relu+conv_trans8,ks2,stride2 \n
//                               conv8,ks3,stride1+norm,leaky_relu+conv3,ks1,stride1
struct RefNet : torch::nn::Module {
    torch::nn::Sequential e0, e1, d0, o0;
    RefNet() {
        namespace nn = torch::nn;
        e0 = nn::Sequential(nn::Conv3d(nn::Conv3dOptions(1, 8, 3).stride(1).padding(1)), nn::InstanceNorm3d(nn::InstanceNorm3dOptions(8).affine(true)),
                            nn::LeakyReLU(nn::LeakyReLUOptions().negative_slope(0.01)));
        e1 = nn::Sequential(nn::Conv3d(nn::Conv3dOptions(8, 16, 3).stride(2).padding(1)), nn::InstanceNorm3d(nn::InstanceNorm3dOptions(16).affine(true)),
                            nn::LeakyReLU(nn::LeakyReLUOptions().negative_slope(0.01)), nn::ConvTranspose3d(nn::ConvTranspose3dOptions(16, 8, 2).stride(2)));
        d0 = nn::Sequential(nn::Conv3d(nn::Conv3dOptions(16, 8, 3).stride(1).padding(1)), nn::InstanceNorm3d(nn::InstanceNorm3dOptions(8).affine(true)),
                            nn::LeakyReLU(nn::LeakyReLUOptions().negative_slope(0.01)));
        o0 = nn::Sequential(nn::Conv3d(nn::Conv3dOptions(8, 3, 1).stride(1).padding(0)));
        register_module("encode0", e0); register_module("encode1", e1); register_module("decode0", d0); register_module("output0", o0);
    }
    torch::Tensor forward(torch::Tensor x) {
        auto s = e0->forward(x);
        auto y = e1->forward(s);
        y = d0->forward(torch::cat({s, y}, 1));
        return o0->forward(y);
    }
};

int main(int argc, char** argv) {
    bool bf16 = argc > 1 && std::string(argv[1]) == "bf16";
    double tol = bf16 ? 4e-2 : 1e-4, gtol = bf16 ? 8e-2 : 2e-4;
    if (!torch::cuda::is_available()) { std::cerr << "needs a GPU" << std::endl; return 2; }
    torch::manual_seed(0);
    const std::string arch = "conv8,ks3,stride1+norm,leaky_relu\nconv16,ks3,stride2+norm,leaky_relu+conv_trans8,ks2,stride2\n"
                             "conv8,ks3,stride1+norm,leaky_relu+conv3,ks1,stride1";
    // DSL errors surface as std::runtime_error with the reference's messages (unet.cpp:66)
    try { UNet3d bad(1, 3, std::string("conv8,ks5\nconv8\nconv8")); REQUIRE(false, "bad DSL accepted"); }
    catch (const std::runtime_error& e) { REQUIRE(std::string(e.what()).find("conv supports only") != std::string::npos, e.what()); }

    auto ref = std::make_shared<RefNet>();
    UNet3d model(1, 3, arch);
    model->engine_dtype = bf16 ? 1 : 0;
    {   // same parameters() order and named_parameters() keys as the reference's module tree
        auto rp = ref->named_parameters(), mp = model->named_parameters();
        REQUIRE(rp.size() == mp.size(), "parameter count");
        torch::NoGradGuard ng;
        for (size_t i = 0; i < rp.size(); ++i) {
            REQUIRE(rp[i].key() == mp[i].key(), "parameter name " + rp[i].key() + " vs " + mp[i].key());
            REQUIRE(rp[i].value().sizes() == mp[i].value().sizes(), "parameter shape " + rp[i].key());
            mp[i].value().copy_(rp[i].value());
        }
    }
    torch::Device dev(torch::kCUDA, 0);
    model->to(dev);                       // torch::nn::Module::to, as train.cpp:940 does
    model->train();
    model->create_optimizer(0.01f);
    REQUIRE(model->device() == dev, "device()");

    auto x = torch::rand({1, 1, 12, 16, 20});
    auto w = torch::randn({1, 3, 12, 16, 20});
    ref->train();
    auto yr = ref->forward(x);
    (yr * w).sum().backward();
    auto outs = model->forward(x.to(dev));
    REQUIRE(outs.size() == 1 && outs[0].sizes() == yr.sizes(), "output shape");
    REQUIRE(rel(outs[0], yr) < tol, "train-mode logits: " + std::to_string(rel(outs[0], yr)));
    (outs[0] * w.to(dev)).sum().backward();
    double gmax = 0;
    for (auto& p : ref->parameters()) gmax = std::max(gmax, p.grad().abs().max().item<double>());
    auto rp = ref->parameters(), mp = model->parameters();
    for (size_t i = 0; i < rp.size(); ++i) {
        REQUIRE(mp[i].grad().defined(), "grad defined");
        double e = (mp[i].grad().to(torch::kCPU) - rp[i].grad()).abs().max().item<double>() / gmax;
        REQUIRE(e < gtol, "grad " + std::to_string(i) + ": " + std::to_string(e));
    }
    // step epilogue exactly as train.cpp:759-766 does it, with the torch optimizer both sides
    auto ropt = torch::optim::SGD(ref->parameters(), torch::optim::SGDOptions(0.01).momentum(0.99).nesterov(true));
    for (auto& p : model->parameters()) p.grad().div_(1);
    torch::nn::utils::clip_grad_norm_(model->parameters(), 12.0);
    torch::nn::utils::clip_grad_norm_(ref->parameters(), 12.0);
    model->optimizer->step(); model->optimizer->zero_grad();
    // second micro-step after zero_grad (grads were set to None): accumulation path
    auto outs2 = model->forward(x.to(dev));
    (outs2[0] * w.to(dev)).sum().backward();
    REQUIRE(model->parameters()[0].grad().defined() && model->parameters()[0].grad().abs().max().item<float>() > 0, "grad after zero_grad");
    // eval: prepare_for_inference + no-grad forward (evaluate.cpp:211-246)
    model->prepare_for_inference(dev);
    {
        torch::NoGradGuard ng;
        auto ye = model->forward(x.to(dev))[0];
        REQUIRE(ye.sizes() == yr.sizes() && torch::isfinite(ye).all().item<bool>(), "eval forward");
    }
    // copy_from / add_gradient_from between replicas (train.cpp:575,757)
    UNet3d replica(1, 3, arch);
    replica->engine_dtype = model->engine_dtype;
    replica->to(dev);
    replica->copy_from(*model);
    REQUIRE(rel(replica->parameters()[0], model->parameters()[0]) == 0.0, "copy_from");
    replica->train();
    auto outs3 = replica->forward(x.to(dev));
    (outs3[0] * w.to(dev)).sum().backward();
    auto before = model->flat_grads.clone();
    model->add_gradient_from(*replica);
    REQUIRE(rel(model->flat_grads, before + replica->flat_grads) < 1e-6, "add_gradient_from");
    // ---- workspace pool (train.cpp:592-594: fresh std::threads every optimizer step; qc.cpp:273-297: 4 workers on one model) ----
    {
        auto reserved = [&]() {
            torch::cuda::synchronize();
            return (long long)c10::hip::HIPCachingAllocator::getDeviceStats(0).reserved_bytes[0].current;
        };
        replica->train();
        auto xd = x.to(dev), wd = w.to(dev);
        std::atomic<int> failures{0};
        auto micro_step = [&]() {   // what a GPU thread of train.cpp does with its replica: forward, loss, backward
            try {
                auto o = replica->forward(xd);
                (o[0] * wd).sum().backward();
            } catch (const std::exception& e) { std::cerr << e.what() << std::endl; ++failures; }
        };
        for (int i = 0; i < 3; ++i) { std::thread t(micro_step); t.join(); }    // warm the allocator and the pool
        const long long r0 = reserved();
        const size_t idle0 = replica->pooled_workspaces();
        for (int i = 0; i < 20; ++i) { std::thread t(micro_step); t.join(); }    // 20 micro-steps, each from a FRESH thread
        REQUIRE(failures == 0, "micro-step from a fresh thread threw");
        REQUIRE(reserved() == r0, "device memory grew under fresh-thread callers: " + std::to_string(reserved() - r0) + " bytes");
        REQUIRE(replica->pooled_workspaces() == idle0 && idle0 >= 1 && idle0 <= 4, "pool size " + std::to_string(replica->pooled_workspaces()));
        // a forward whose outputs are dropped without backward still returns its lease
        { auto o = replica->forward(xd); }
        REQUIRE(replica->pooled_workspaces() == idle0, "lease of an unused training forward was not returned");
        // 4 concurrent eval forwards on ONE model (qc.cpp:273-297) agree with a serial one, and the pool stays bounded
        model->eval();
        torch::Tensor serial;
        { torch::NoGradGuard ng; serial = model->forward(xd)[0].clone(); }
        std::vector<torch::Tensor> got(4);
        std::vector<std::thread> th;
        for (int k = 0; k < 4; ++k)
            th.emplace_back([&, k]() {
                try {
                    torch::NoGradGuard ng;
                    torch::Tensor last;
                    for (int r = 0; r < 5; ++r) last = model->forward(xd)[0];
                    got[k] = last.clone();
                } catch (const std::exception& e) { std::cerr << e.what() << std::endl; ++failures; }
            });
        for (auto& t : th) t.join();
        torch::cuda::synchronize();
        REQUIRE(failures == 0, "concurrent eval forward threw");
        for (int k = 0; k < 4; ++k) REQUIRE(got[k].defined() && torch::equal(got[k], serial), "concurrent eval forward " + std::to_string(k) + " differs");
        REQUIRE(model->pooled_workspaces() <= 4, "pool exceeds its bound");
    }
    // ---- RCCL under the C ABI, driven from the C++ host: a one-rank communicator must leave the step unchanged ----
    {
        char id[UNET_COMM_ID_BYTES];
        unet_comm* comm = nullptr;
        REQUIRE(unet_comm_unique_id(id) == 0, std::string("unet_comm_unique_id: ") + unet_last_error());
        REQUIRE(unet_comm_create(0, 1, id, 0, &comm) == 0, std::string("unet_comm_create: ") + unet_last_error());
        UNet3d a(1, 3, arch), b(1, 3, arch);
        a->engine_dtype = b->engine_dtype = model->engine_dtype;
        a->to(dev); b->to(dev);
        b->copy_from(*a);
        a->train(); b->train();
        b->attach_comm(comm);
        b->broadcast_parameters(0);
        auto xd = x.to(dev);
        auto tgt = torch::randint(0, 3, {1, 12, 16, 20}, torch::TensorOptions().dtype(torch::kLong).device(dev));
        for (int stp = 0; stp < 2; ++stp) {
            auto la = a->loss_and_backward(xd, tgt, true, true, true);
            a->sgd_step(0.01f, 1.0f);
            auto lb = b->loss_and_backward_overlapped(xd, tgt, true, true, true);     // bucketed backward + asynchronous all-reduces
            b->allreduce_gradients();                                                // the rest + join (train.cpp:756-757)
            b->sgd_step(0.01f, 1.0f);
            REQUIRE(torch::equal(la, lb), "losses with a one-rank communicator");
        }
        torch::cuda::synchronize();
        REQUIRE(torch::equal(a->flat_params, b->flat_params), "parameters with a one-rank communicator differ from the no-collective run");
        b->attach_comm(nullptr);
        REQUIRE(unet_comm_destroy(comm) == 0, "unet_comm_destroy");
    }
    // ---- the reference's own multi-GPU model: ONE process, one communicator per device created together, collectives issued as a
    // group from one thread (train.cpp:592-600 starts a std::thread per GPU; :756-757 / :573-579 are the reduce and the broadcast).
    // The test box has one GPU, so n = 1 on device 0: unet_comm_create_all + unet_allreduce_grads_all + broadcast + join must leave a
    // step bit-identical to the run without any collective, and null entries are rejected before RCCL sees them. ----
    {
        int devs[1] = {0};
        unet_comm* comms[1] = {nullptr};
        REQUIRE(unet_comm_create_all(1, devs, comms) == 0 && comms[0], std::string("unet_comm_create_all: ") + unet_last_error());
        int rk = -1, wd = -1;
        REQUIRE(unet_comm_rank(comms[0], &rk, &wd) == 0 && rk == 0 && wd == 1, "unet_comm_rank of a create_all communicator");
        UNet3d a(1, 3, arch), b(1, 3, arch);
        a->engine_dtype = b->engine_dtype = model->engine_dtype;
        a->to(dev); b->to(dev);
        b->copy_from(*a);
        a->train(); b->train();
        auto xd = x.to(dev);
        auto tgt = torch::randint(0, 3, {1, 12, 16, 20}, torch::TensorOptions().dtype(torch::kLong).device(dev));
        void* st = (void*)c10::hip::getCurrentHIPStream(0).stream();
        {   // broadcast of the parameters from "rank" 0 (train.cpp:573-579), then join
            REQUIRE(unet_comm_broadcast(comms[0], b->flat_params.data_ptr<float>(), b->flat_params.numel(), 0, st) == 0, unet_last_error());
            REQUIRE(unet_comm_join(comms[0], st) == 0, unet_last_error());
        }
        for (int stp = 0; stp < 2; ++stp) {
            auto la = a->loss_and_backward(xd, tgt, true, true, true);
            a->sgd_step(0.01f, 1.0f);
            auto lb = b->loss_and_backward(xd, tgt, true, true, true);
            float* flats[1] = {b->flat_grads.data_ptr<float>()};
            void* streams[1] = {st};
            REQUIRE(unet_allreduce_grads_all(comms, 1, flats, 0, b->flat_grads.numel(), streams) == 0, std::string("unet_allreduce_grads_all: ") + unet_last_error());
            REQUIRE(unet_comm_join(comms[0], st) == 0, unet_last_error());
            b->sgd_step(0.01f, 1.0f);
            REQUIRE(torch::equal(la, lb), "losses with the in-process communicator group");
        }
        torch::cuda::synchronize();
        REQUIRE(torch::equal(a->flat_params, b->flat_params), "parameters with the in-process communicator group differ from the no-collective run");
        {   // a null communicator / buffer in the group is an argument error, not a crash inside RCCL
            unet_comm* bad[1] = {nullptr};
            float* flats[1] = {b->flat_grads.data_ptr<float>()};
            void* streams[1] = {st};
            REQUIRE(unet_allreduce_grads_all(bad, 1, flats, 0, 16, streams) != 0, "null communicator accepted");
            float* nof[1] = {nullptr};
            REQUIRE(unet_allreduce_grads_all(comms, 1, nof, 0, 16, streams) != 0, "null buffer accepted");
        }
        REQUIRE(unet_comm_destroy(comms[0]) == 0, "unet_comm_destroy");
    }
    // ---- two replicas driven by two host threads at once (train.cpp:592-600: a std::thread per GPU, each on its own replica; here
    // both replicas live on the one device): thread B runs forward + backward micro-steps on `rep` while the main thread trains
    // `root`; then the reduce-to-root and the broadcast of train.cpp:756-757,573-579.  The sum must equal the two gradients computed
    // one after the other, bit for bit (the engine's reductions have a fixed order; the streams differ per thread). ----
    {
        UNet3d root(1, 3, arch), rep(1, 3, arch), solo(1, 3, arch);
        for (auto* m : {&root, &rep, &solo}) { (*m)->engine_dtype = model->engine_dtype; (*m)->to(dev); (*m)->train(); }
        rep->copy_from(*root); solo->copy_from(*root);
        auto xa = x.to(dev), xb = torch::rand({1, 1, 12, 16, 20}).to(dev);
        auto ta = torch::randint(0, 3, {1, 12, 16, 20}, torch::TensorOptions().dtype(torch::kLong).device(dev));
        auto tb = torch::randint(0, 3, {1, 12, 16, 20}, torch::TensorOptions().dtype(torch::kLong).device(dev));
        torch::cuda::synchronize();
        std::atomic<int> failures{0};
        for (int stp = 0; stp < 3; ++stp) {
            std::thread tb_thread([&]() {
                try {
                    // a thread's own stream, as a replica on another GPU would have (the current stream is per thread; the main thread
                    // synchronises the device after the join)
                    c10::hip::setCurrentHIPStream(c10::hip::getStreamFromPool(false, 0));
                    rep->loss_and_backward(xb, tb, true, true, true);
                } catch (const std::exception& e) { std::cerr << e.what() << std::endl; ++failures; }
            });
            root->loss_and_backward(xa, ta, true, true, true);
            tb_thread.join();
            torch::cuda::synchronize();
            root->add_gradient_from(*rep);          // train.cpp:756-757
            rep->flat_grads.zero_();
            root->sgd_step(0.01f, 0.5f);
            rep->copy_from(*root);                  // train.cpp:573-579
            // the same step on one model, serially
            solo->loss_and_backward(xa, ta, true, true, true);
            auto ga = solo->flat_grads.clone();
            solo->flat_grads.zero_();
            solo->loss_and_backward(xb, tb, true, true, true);
            solo->flat_grads.copy_(ga + solo->flat_grads);
            solo->sgd_step(0.01f, 0.5f);
        }
        torch::cuda::synchronize();
        REQUIRE(failures == 0, "replica thread threw");
        REQUIRE(torch::equal(root->flat_params, solo->flat_params), "two replicas on two threads differ from the serial run");
        REQUIRE(torch::equal(root->flat_params, rep->flat_params), "copy_from after the step");
    }
    // ---- resume (train.cpp:787: torch::save(*optimizer, path + ".opt"); :945-957: torch::load on restart): two steps, checkpoint
    // (network file + optimizer), a fresh model that loads both, two more steps == four uninterrupted steps, bit for bit ----
    {
        UNet3d a(1, 3, arch), b(1, 3, arch);
        a->engine_dtype = b->engine_dtype = model->engine_dtype;
        a->to(dev); b->to(dev);
        b->copy_from(*a);
        a->train(); b->train();
        a->create_optimizer(0.01f); b->create_optimizer(0.01f);
        auto xd = x.to(dev);
        auto tgt = torch::randint(0, 3, {1, 12, 16, 20}, torch::TensorOptions().dtype(torch::kLong).device(dev));
        for (int stp = 0; stp < 4; ++stp) { a->loss_and_backward(xd, tgt, true, true, true); a->sgd_step(0.01f, 1.0f); }
        for (int stp = 0; stp < 2; ++stp) { b->loss_and_backward(xd, tgt, true, true, true); b->sgd_step(0.01f, 1.0f); }
        const std::string base = std::string("/tmp/test_unet_hpp_resume_") + (bf16 ? "bf16" : "fp32") + "_" + std::to_string((long long)getpid());
        REQUIRE(save_to_file(b, (base + ".nz").c_str()), "save_to_file");
        REQUIRE(b->save_optimizer(base + ".nz.opt"), "save_optimizer: " + b->error_msg);
        UNet3d c;
        REQUIRE(load_from_file(c, (base + ".nz").c_str()), "load_from_file");
        c->engine_dtype = model->engine_dtype;
        c->to(dev); c->train();
        c->create_optimizer(0.01f);
        REQUIRE(c->load_optimizer(base + ".nz.opt"), "load_optimizer: " + c->error_msg);
        for (int stp = 0; stp < 2; ++stp) { c->loss_and_backward(xd, tgt, true, true, true); c->sgd_step(0.01f, 1.0f); }
        torch::cuda::synchronize();
        REQUIRE(torch::equal(a->flat_params, c->flat_params), "resumed run differs from the uninterrupted one");
        // and WITHOUT the optimizer file the momentum is lost: the test would not notice a save that wrote nothing otherwise
        UNet3d d;
        REQUIRE(load_from_file(d, (base + ".nz").c_str()), "load_from_file (2)");
        d->engine_dtype = model->engine_dtype;
        d->to(dev); d->train(); d->create_optimizer(0.01f);
        for (int stp = 0; stp < 2; ++stp) { d->loss_and_backward(xd, tgt, true, true, true); d->sgd_step(0.01f, 1.0f); }
        torch::cuda::synchronize();
        REQUIRE(!torch::equal(a->flat_params, d->flat_params), "momentum made no difference: the resume test is vacuous");
        // the plain calls train.cpp makes work too: torch::save(*optimizer) after a fused step carries the fused momentum
        torch::save(*(c->optimizer), base + ".2.opt");
        UNet3d e2(1, 3, arch);
        e2->engine_dtype = model->engine_dtype;
        e2->to(dev); e2->train(); e2->create_optimizer(0.01f);
        torch::load(*(e2->optimizer), base + ".2.opt");
        e2->bind_optimizer_state();
        e2->copy_from(*c);
        c->loss_and_backward(xd, tgt, true, true, true); c->sgd_step(0.01f, 1.0f);
        e2->loss_and_backward(xd, tgt, true, true, true); e2->sgd_step(0.01f, 1.0f);
        torch::cuda::synchronize();
        REQUIRE(torch::equal(c->flat_params, e2->flat_params), "torch::save / torch::load of *optimizer lost the fused momentum");
        std::remove((base + ".nz").c_str()); std::remove((base + ".nz.opt").c_str()); std::remove((base + ".2.opt").c_str());
    }
    std::cout << model->get_info();
    std::cout << "OK " << (bf16 ? "bf16" : "fp32") << " logits rel " << rel(outs[0], yr) << std::endl;
    return 0;
}
