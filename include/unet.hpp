// Drop-in replacement for the reference's unet.hpp (unet.hpp:1-73): the same `UNet3dImpl` / `UNet3d`
// operator surface -- public fields, method names, signatures and error behaviour -- over libunet_hip.so.
// train.cpp, evaluate.cpp, qc.cpp, main.cpp and the Qt front-end compile against it unchanged; the
// implementation (unet-studio_amd/csrc/unet_host.cpp) replaces unet.cpp and drives the HIP engine through the
// C ABI of unet_hip.h instead of building torch::nn conv/norm modules.
#ifndef UNET_HPP
#define UNET_HPP
#ifdef QT_CORE_LIB
    #undef slots
#endif
#include <torch/torch.h>
#ifdef QT_CORE_LIB
    #define slots Q_SLOTS
#endif
#include <array>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#if __has_include("TIPL/tipl.hpp")
#include "TIPL/tipl.hpp"
namespace unet_types { typedef tipl::vector<3> vector3; typedef tipl::shape<3> shape3; }
#else
// TIPL is not vendored with the reference; without it the two geometry fields fall back to plain PODs with the
// operations this class needs (indexing, streaming).  With TIPL on the include path the TIPL types are used.
namespace unet_types {
struct vector3 {
    float v[3];
    vector3(float x = 1.f, float y = 1.f, float z = 1.f) : v{x, y, z} {}
    float& operator[](size_t i) { return v[i]; }
    const float& operator[](size_t i) const { return v[i]; }
};
struct shape3 {
    unsigned v[3];
    shape3(unsigned x = 0, unsigned y = 0, unsigned z = 0) : v{x, y, z} {}
    unsigned& operator[](size_t i) { return v[i]; }
    const unsigned& operator[](size_t i) const { return v[i]; }
    size_t size() const { return (size_t)v[0] * v[1] * v[2]; }
};
inline std::ostream& operator<<(std::ostream& o, const vector3& a) { return o << a[0] << " " << a[1] << " " << a[2]; }
inline std::ostream& operator<<(std::ostream& o, const shape3& a) { return o << a[0] << " " << a[1] << " " << a[2]; }
}  // namespace unet_types
#endif

struct unet_plan;
struct unet_comm;

struct UNet3dImpl : torch::nn::Module
{
public:
    int in_count = 1;
    int out_count = 1;
    std::string architecture,preproc,postproc,orientation,fov_strategy,error_msg;
public:
    std::shared_ptr<torch::optim::SGD> optimizer;

    std::vector<float> testing_errors,training_errors;
    std::vector<unsigned int> single_component_label;
    mutable std::mutex error_mutex;
    auto get_training_errors(void) const
    {
        std::scoped_lock<std::mutex> lock(error_mutex);
        return std::vector<float>(training_errors);
    }
    auto get_testing_errors(void) const
    {
        std::scoped_lock<std::mutex> lock(error_mutex);
        return std::vector<float>(testing_errors);
    }

public:
    unet_types::vector3 voxel_size = {1.0f,1.0f,1.0f};
    unet_types::shape3 dim = {192,224,192};
    // kept for source compatibility (the reference exposes them); the layers live in the lowered HIP plan
    std::deque<torch::nn::Sequential> encoding,decoding,decoding_tail;
    std::vector<torch::nn::Sequential> output;
    int create_layer(torch::nn::Sequential& layers,const std::string& def, int in_c);
public:
    std::string get_info(void) const;
public:
    UNet3dImpl(void){}
    UNet3dImpl(int32_t in_count_,int32_t out_count_,std::string);
    ~UNet3dImpl(void);
    void copy_from(const UNet3dImpl& r);
    void add_gradient_from(const UNet3dImpl& r);
    void create_optimizer(float learning_rate);
public:
    std::vector<torch::Tensor> forward(torch::Tensor inputTensor);

    void set_requires_grad(bool req)
    {
        for (auto& p : parameters())
            p.set_requires_grad(req);
    }
    virtual void train(bool on = true) override
    {
        set_requires_grad(on);
        torch::nn::Module::train(on);
    }
    void print_layers(void);
    torch::Device device(void) const
    {
        return parameters().size() && parameters()[0].defined() ? parameters()[0].device() : torch::kCPU;
    }
    void prepare_for_inference(const torch::Device& device);

public: // ---- engine side (not in the reference) ----
    size_t pooled_workspaces(void) const;   // idle workspaces held by the pool (tests: bounded under fresh-thread callers)
    int engine_dtype = 1;                  // UNET_DTYPE_BF16 (0 = fp32 parity configuration)
    torch::Tensor flat_params, flat_grads; // fp32, parameters() order; the registered parameters are views
    void to_device(const torch::Device& device);   // what to(device) means for the flat storage
    // fused micro-step pieces (train.cpp:634-706 and 759-766 without autograd), optional for callers
    torch::Tensor loss_and_backward(torch::Tensor input, torch::Tensor target_int64, bool ce, bool dice, bool mse, int collapse_before = 0);
    void sgd_step(float lr, float grad_scale, float clip_norm = 12.0f);
    // Resume path (train.cpp:787, :945-957).  The momentum of the fused update and the momentum_buffer tensors in *optimizer's state
    // are the same memory once bind_optimizer_state() has run (sgd_step / save_optimizer / load_optimizer call it), so
    // torch::save(*optimizer, path) and torch::load(*optimizer, path) + bind_optimizer_state() work as train.cpp writes them;
    // save_optimizer / load_optimizer are those two calls with the binding included (false + error_msg on failure).
    void bind_optimizer_state(void);
    bool save_optimizer(const std::string& file_name);
    bool load_optimizer(const std::string& file_name);
    // Data parallel over RCCL (one process per GPU; unet_hip.h unet_comm_*).  With a communicator attached:
    //   broadcast_parameters(0)   once at start             replaces  other_models[i]->copy_from(*model)   train.cpp:573-579
    //   allreduce_gradients()     after the micro-steps     replaces  model->add_gradient_from(*replica)   train.cpp:756-757
    //   loss_and_backward(..., last_micro_step = true) starts the all-reduce of every finished gradient bucket under the rest of
    //   the backward; allreduce_gradients() then only reduces what is left and joins.  Every rank runs the same sgd_step.
    void attach_comm(struct unet_comm* comm) { comm_ = comm; reduced_from_ = -1; }
    void broadcast_parameters(int root = 0);
    void broadcast_buffers(int root = 0);  // BatchNorm running statistics follow the root (copy_from, unet.cpp:207-215); allreduce_gradients() does it each step
    void allreduce_gradients(void);
    torch::Tensor loss_and_backward_overlapped(torch::Tensor input, torch::Tensor target_int64, bool ce, bool dice, bool mse, int collapse_before = 0);
private:
    friend struct UNetForwardFn;
    std::vector<torch::Tensor> params_, buffers_;
    std::map<std::array<int64_t,3>, unet_plan*> plans_;
    // Workspaces are leased per call from a small pool and go back when the call's last user is done (the no-grad forward on
    // return, the training forward when its autograd node has run backward or is dropped): train.cpp:592-594 starts fresh
    // std::threads every optimizer step, so nothing may be keyed by thread id; qc.cpp:273-297 runs up to 4 forwards at once.
    struct WorkspacePool;
    std::shared_ptr<WorkspacePool> ws_pool_;
    std::mutex plans_mutex_;
    torch::Tensor trigger_, momentum_, scratch_;
    struct unet_comm* comm_ = nullptr;
    int64_t reduced_from_ = -1;            // flat gradient elements [reduced_from_, end) are already being all-reduced (this step)
    unet_plan* plan_for(int64_t d,int64_t h,int64_t w);
    torch::Tensor workspace_for(unet_plan* plan);   // a lease: returns itself to the pool when the last reference goes away
    void bind_views(void);
    void ensure_flat(void);
    void rebind_grads(void);
    std::vector<torch::Tensor> run_forward(unet_plan* plan, torch::Tensor ws, torch::Tensor x, int mode);
    void run_backward(unet_plan* plan, torch::Tensor ws, const std::vector<torch::Tensor>& grad_outs);
};
TORCH_MODULE_IMPL(UNet3d, UNet3dImpl);

// `.nz` network files (the reference declares these in train.hpp:32-33 and defines them in main.cpp:157-233 over TIPL's gz_mat
// container; here: unet-studio_amd/csrc/nz_io.cpp, gzip + MATLAB Level-4 records, plain float tensors)
bool save_to_file(UNet3d& model,const char* file_name);
bool load_from_file(UNet3d& model,const char* file_name);


#endif// UNET_HPP
