// Replacement for the reference's unet.cpp: the UNet3dImpl member functions of include/unet.hpp, driving the
// HIP engine through the C ABI (include/unet_hip.h).  libtorch is used for tensors, autograd bookkeeping and the
// optimizer object only; no torch::nn conv/norm module is built.  Each function cites the reference lines it replaces.
#include "../../include/unet.hpp"
#include "../../include/unet_hip.h"

#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime_api.h>

#include <functional>
#include <sstream>
#include <thread>

namespace {

void check(int rc) { if (rc) throw std::runtime_error(unet_last_error()); }

void* stream_of(const torch::Device& d) {
    return d.is_cuda() ? (void*)c10::hip::getCurrentHIPStream(d.index()).stream() : nullptr;
}

// plain module used to rebuild the reference's module tree ("encode0" -> "3" -> "weight"), so that
// named_parameters() yields the same keys as the torch::nn::Sequential tree of unet.cpp:130,160-164
struct Holder : torch::nn::Module {};

}  // namespace

// Workspace pool.  An entry = the device buffer + an event recorded on the stream of its last user when it came back; the next
// lease makes its own stream wait for that event, so a buffer can move between host threads and streams.  At most MAX_IDLE idle
// buffers per plan are kept (qc.cpp:263: up to 4 workers); a 128^3 bf16 workspace is ~1 GB, so "one per thread id ever seen"
// (the first version) grew without bound under train.cpp:592-594's fresh std::threads.
struct UNet3dImpl::WorkspacePool {
    struct Entry { torch::Tensor buf; hipEvent_t ev = nullptr; };
    static constexpr size_t MAX_IDLE = 4;
    std::mutex m;
    std::map<unet_plan*, std::vector<Entry>> idle;
    ~WorkspacePool() { clear(); }
    void clear() {
        std::scoped_lock<std::mutex> lock(m);
        for (auto& kv : idle)
            for (auto& e : kv.second)
                if (e.ev) (void)hipEventDestroy(e.ev);
        idle.clear();
    }
    size_t count() {
        std::scoped_lock<std::mutex> lock(m);
        size_t n = 0;
        for (auto& kv : idle) n += kv.second.size();
        return n;
    }
};

// ---- unet.cpp:103-166: constructor -------------------------------------------------------------------------
UNet3dImpl::UNet3dImpl(int32_t in_count_, int32_t out_count_, std::string architecture_)
    : in_count(in_count_), out_count(out_count_), architecture(architecture_)
{
    fov_strategy = "align_top";
    preproc = "";
    postproc = "softmax+create_mask+argmax";
    // a probe plan validates the DSL (throws std::runtime_error with the reference's messages) and gives the
    // size-independent facts: parameter order, shapes, names, default-init bounds
    unet_plan* probe = nullptr;
    check(unet_plan_create(architecture_.c_str(), in_count_, out_count_, 1024, 1024, 1024, engine_dtype, 0, UNET_IMPL_AUTO, &probe));
    int n = 0;
    unet_plan_param_count(probe, &n);
    std::vector<std::vector<int64_t>> shapes(n);
    std::vector<std::string> names(n);
    std::vector<int64_t> fan(n);
    std::vector<int> isnw(n);
    int64_t total = 0;
    for (int i = 0; i < n; ++i) {
        int64_t d[5]; int nd = 0; char nm[128];
        check(unet_plan_param_shape(probe, i, d, &nd));
        check(unet_plan_param_name(probe, i, nm, sizeof(nm)));
        check(unet_plan_param_fan_in(probe, i, &fan[i], &isnw[i]));
        shapes[i].assign(d, d + nd);
        names[i] = nm;
        int64_t e = 1;
        for (auto v : shapes[i]) e *= v;
        total += e;
    }
    flat_params = torch::zeros({total});
    flat_grads = torch::zeros({total});
    // libtorch default init (Conv3d / ConvTranspose3d reset_parameters: kaiming_uniform_(a = sqrt(5)) == U(+-1/sqrt(fan_in)),
    // bias U(+-1/sqrt(fan_in)); norm weight 1, bias 0)
    int64_t off = 0;
    std::map<std::string, std::shared_ptr<Holder>> tops, subs;
    for (int i = 0; i < n; ++i) {
        int64_t e = 1;
        for (auto v : shapes[i]) e *= v;
        auto view = torch::from_blob(flat_params.data_ptr<float>() + off, shapes[i], torch::TensorOptions().dtype(torch::kFloat32));
        if (fan[i] > 0) view.uniform_(-1.0 / std::sqrt((double)fan[i]), 1.0 / std::sqrt((double)fan[i]));
        else view.fill_(isnw[i] ? 1.0f : 0.0f);
        off += e;
        // "encode0.3.weight" -> module "encode0" / module "3" / parameter "weight"
        size_t p1 = names[i].find('.'), p2 = names[i].rfind('.');
        std::string top = names[i].substr(0, p1), mid = names[i].substr(p1 + 1, p2 - p1 - 1), leaf = names[i].substr(p2 + 1);
        if (!tops.count(top)) tops[top] = register_module(top, std::make_shared<Holder>());
        std::string key = top + "." + mid;
        if (!subs.count(key)) subs[key] = tops[top]->register_module(mid, std::make_shared<Holder>());
        params_.push_back(subs[key]->register_parameter(leaf, view, true));
    }
    int nb = 0;
    unet_plan_buffer_count(probe, &nb);
    for (int i = 0; i < nb; ++i) {
        int64_t ne = 0;
        unet_plan_buffer_shape(probe, i, &ne);
        // running_mean (0) / running_var (1) pairs, as BatchNorm3d registers them
        buffers_.push_back(register_buffer("bnorm" + std::to_string(i / 2) + (i % 2 ? "_running_var" : "_running_mean"),
                                           i % 2 ? torch::ones({ne}) : torch::zeros({ne})));
    }
    unet_plan_destroy(probe);
    trigger_ = torch::zeros({1}, torch::requires_grad(true));
    bind_views();
}

UNet3dImpl::~UNet3dImpl(void)
{
    for (auto& kv : plans_) unet_plan_destroy(kv.second);
}

// parameters and their .grad are views into the two flat buffers (one all-reduce / one SGD launch over everything)
void UNet3dImpl::bind_views(void)
{
    // aliases made with from_blob, not autograd views: optimizer->zero_grad() calls grad.detach_(), which views refuse
    int64_t off = 0;
    auto opts = torch::TensorOptions().dtype(torch::kFloat32).device(flat_params.device());
    for (auto& p : params_) {
        int64_t e = p.numel();
        torch::NoGradGuard ng;
        p.set_data(torch::from_blob(flat_params.data_ptr<float>() + off, p.sizes(), opts));
        p.mutable_grad() = torch::from_blob(flat_grads.data_ptr<float>() + off, p.sizes(), opts);
        off += e;
    }
}

void UNet3dImpl::to_device(const torch::Device& device)
{
    if (flat_params.defined() && flat_params.device() != device) {
        flat_params = flat_params.to(device);
        flat_grads = flat_grads.to(device);
        for (auto& b : buffers_) b.set_data(b.to(device));
        bind_views();
    }
    ensure_flat();
}

// torch::nn::Module::to(device) (train.cpp:940,968) re-homes every parameter into storage of its own: gather them back
// into the flat buffers (on whatever device they are now) before the engine touches them
void UNet3dImpl::ensure_flat(void)
{
    if (params_.empty()) return;
    auto dev = params_[0].device();
    bool ok = flat_params.defined() && flat_params.device() == dev && params_[0].data_ptr() == flat_params.data_ptr();
    if (!ok) {
        torch::NoGradGuard ng;
        std::vector<torch::Tensor> ps, gs;
        for (auto& p : params_) {
            ps.push_back(p.detach().to(dev).reshape({-1}).to(torch::kFloat32));
            gs.push_back(p.grad().defined() ? p.grad().detach().to(dev).reshape({-1}).to(torch::kFloat32) : torch::zeros({p.numel()}, ps.back().options()));
        }
        flat_params = torch::cat(ps).contiguous();
        flat_grads = torch::cat(gs).contiguous();
        bind_views();
    }
    if (!trigger_.defined() || trigger_.device() != dev) {
        trigger_ = torch::zeros({1}, torch::TensorOptions().device(dev).requires_grad(true));
        if (momentum_.defined()) {
            // the optimizer state's momentum_buffer tensors alias momentum_ (bind_optimizer_state): drop the aliases before the
            // buffer they point into is replaced; the next bind re-aliases the moved buffer without copying from freed memory
            if (optimizer) {
                const char* lo = (const char*)momentum_.data_ptr();
                const char* hi = lo + momentum_.numel() * sizeof(float);
                for (auto& kv : optimizer->state()) {
                    auto& ps = static_cast<torch::optim::SGDParamState&>(*kv.second);
                    const char* q = ps.momentum_buffer().defined() ? (const char*)ps.momentum_buffer().data_ptr() : nullptr;
                    if (q && q >= lo && q < hi) ps.momentum_buffer(torch::Tensor());
                }
            }
            momentum_ = momentum_.to(dev);
        }
        scratch_ = torch::Tensor();
        std::scoped_lock<std::mutex> lock(plans_mutex_);
        for (auto& kv : plans_) unet_plan_destroy(kv.second);   // plans are bound to a device
        plans_.clear();
        if (ws_pool_) ws_pool_->clear();
    }
    for (auto& b : buffers_)
        if (b.device() != dev) b.set_data(b.to(dev));
}

// optimizer->zero_grad() (train.cpp:766) sets .grad to None in libtorch 2.x: a missing gradient means zero, so
// clear that slice of the flat buffer and point .grad at it again before the engine accumulates into it
void UNet3dImpl::rebind_grads(void)
{
    torch::NoGradGuard ng;
    int64_t off = 0;
    auto opts = torch::TensorOptions().dtype(torch::kFloat32).device(flat_grads.device());
    for (auto& p : params_) {
        int64_t e = p.numel();
        float* slot = flat_grads.data_ptr<float>() + off;
        if (!p.grad().defined() || p.grad().data_ptr() != (void*)slot) {
            auto alias = torch::from_blob(slot, p.sizes(), opts);
            if (p.grad().defined()) alias.copy_(p.grad()); else alias.zero_();
            p.mutable_grad() = alias;
        }
        off += e;
    }
}

int UNet3dImpl::create_layer(torch::nn::Sequential&, const std::string& def, int)
{
    throw std::runtime_error("create_layer(" + def + "): layers are lowered into the HIP plan by unet_plan_create");
}

unet_plan* UNet3dImpl::plan_for(int64_t d, int64_t h, int64_t w)
{
    std::scoped_lock<std::mutex> lock(plans_mutex_);
    std::array<int64_t, 3> key = {d, h, w};
    auto it = plans_.find(key);
    if (it != plans_.end()) return it->second;
    unet_plan* p = nullptr;
    auto dev = device();
    check(unet_plan_create(architecture.c_str(), in_count, out_count, (int)d, (int)h, (int)w, engine_dtype,
                           dev.is_cuda() ? dev.index() : 0, UNET_IMPL_AUTO, &p));
    plans_[key] = p;
    return p;
}

size_t UNet3dImpl::pooled_workspaces(void) const { return ws_pool_ ? ws_pool_->count() : 0; }

torch::Tensor UNet3dImpl::workspace_for(unet_plan* plan)
{
    auto dev = device();
    std::shared_ptr<WorkspacePool> pool;
    {   // created and read under the lock (no unsynchronised first read of the shared_ptr)
        std::scoped_lock<std::mutex> lock(plans_mutex_);
        if (!ws_pool_) ws_pool_ = std::make_shared<WorkspacePool>();
        pool = ws_pool_;
    }
    WorkspacePool::Entry e;
    {
        std::scoped_lock<std::mutex> lock(pool->m);
        auto& v = pool->idle[plan];
        if (!v.empty()) { e = v.back(); v.pop_back(); }
    }
    c10::DeviceGuard guard(dev);
    // the stream this lease's kernels are enqueued on: the forward runs on it now, and autograd runs the node's backward on the
    // forward's stream whatever thread executes it -- so this, not "the current stream of whichever thread drops the last
    // reference", is where the buffer's last use is ordered
    void* const lease_stream = stream_of(dev);
    if (e.buf.defined()) {
        if (e.ev) (void)hipStreamWaitEvent((hipStream_t)lease_stream, e.ev, 0);   // the previous user's kernels come first
    } else {
        size_t bytes = 0;
        unet_plan_workspace_bytes(plan, &bytes);
        e.buf = torch::empty({(int64_t)bytes}, torch::TensorOptions().dtype(torch::kUInt8).device(dev));
        if (dev.is_cuda() && hipEventCreateWithFlags(&e.ev, hipEventDisableTiming) != hipSuccess) e.ev = nullptr;
    }
    // the lease aliases the pooled buffer; its deleter (run when the last reference dies: end of a no-grad forward, or the
    // autograd node's saved data after backward / when the outputs are dropped) hands the buffer back
    std::weak_ptr<WorkspacePool> wp = pool;
    auto back = [wp, e, plan, dev, lease_stream](void*) mutable {
        auto p = wp.lock();
        if (e.ev && dev.is_cuda()) {
            c10::DeviceGuard g(dev);
            (void)hipEventRecord(e.ev, (hipStream_t)lease_stream);
        }
        if (p) {
            std::scoped_lock<std::mutex> lock(p->m);
            auto& v = p->idle[plan];
            if (v.size() < WorkspacePool::MAX_IDLE) { v.push_back(e); return; }
        }
        if (e.ev) (void)hipEventDestroy(e.ev);   // pool gone or full: the buffer goes back to torch's allocator with `e`
    };
    return torch::from_blob(e.buf.data_ptr(), e.buf.sizes(), back, e.buf.options());
}

std::vector<torch::Tensor> UNet3dImpl::run_forward(unet_plan* plan, torch::Tensor ws, torch::Tensor x, int mode)
{
    std::vector<const float*> pp;
    for (auto& p : params_) pp.push_back(p.data_ptr<float>());
    std::vector<float*> bp;
    for (auto& b : buffers_) bp.push_back(b.data_ptr<float>());
    int nl = 0;
    unet_plan_output_count(plan, &nl);
    std::vector<torch::Tensor> outs(nl);
    std::vector<float*> op(nl, nullptr);
    for (int l = 0; l < nl; ++l) {
        int64_t d[5];
        unet_plan_output_shape(plan, l, d);
        if (d[1]) {
            outs[l] = torch::empty({d[0], d[1], d[2], d[3], d[4]}, torch::TensorOptions().dtype(torch::kFloat32).device(x.device()));
            op[l] = outs[l].data_ptr<float>();
        }
    }
    check(unet_forward(plan, pp.data(), bp.empty() ? nullptr : bp.data(), x.data_ptr<float>(), op.data(), ws.data_ptr(), mode,
                       stream_of(x.device())));
    return outs;
}

void UNet3dImpl::run_backward(unet_plan* plan, torch::Tensor ws, const std::vector<torch::Tensor>& grad_outs)
{
    rebind_grads();
    std::vector<const float*> pp, go;
    std::vector<float*> gp;
    std::vector<torch::Tensor> keep;
    int64_t off = 0;
    for (auto& p : params_) {
        pp.push_back(p.data_ptr<float>());
        gp.push_back(flat_grads.data_ptr<float>() + off);   // .grad accumulates across micro-steps (train.cpp:604-606,706)
        off += p.numel();
    }
    for (auto& t : grad_outs) {
        if (t.defined()) { keep.push_back(t.contiguous()); go.push_back(keep.back().data_ptr<float>()); }
        else go.push_back(nullptr);
    }
    check(unet_backward(plan, pp.data(), go.data(), gp.data(), nullptr, ws.data_ptr(), stream_of(flat_params.device())));
}

struct UNetForwardFn : torch::autograd::Function<UNetForwardFn> {
    static torch::autograd::variable_list forward(torch::autograd::AutogradContext* ctx, torch::Tensor x, torch::Tensor trigger,
                                                  int64_t self, int64_t plan, torch::Tensor ws) {
        (void)trigger;
        auto* m = reinterpret_cast<UNet3dImpl*>(self);
        auto outs = m->run_forward(reinterpret_cast<unet_plan*>(plan), ws, x, 1);
        ctx->saved_data["self"] = self;
        ctx->saved_data["plan"] = plan;
        ctx->saved_data["ws"] = ws;
        return outs;
    }
    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
        auto* m = reinterpret_cast<UNet3dImpl*>(ctx->saved_data["self"].toInt());
        m->run_backward(reinterpret_cast<unet_plan*>(ctx->saved_data["plan"].toInt()), ctx->saved_data["ws"].toTensor(), grads);
        return {torch::Tensor(), torch::Tensor(), torch::Tensor(), torch::Tensor(), torch::Tensor()};
    }
};

// ---- unet.cpp:168-193: forward ------------------------------------------------------------------------------
std::vector<torch::Tensor> UNet3dImpl::forward(torch::Tensor inputTensor)
{
    if (inputTensor.dim() != 5 || inputTensor.size(0) != 1 || inputTensor.size(1) != in_count)
        throw std::runtime_error("UNet3d::forward expects a {1,in_count,D,H,W} tensor");
    ensure_flat();
    if (inputTensor.device() != device())
        throw std::runtime_error("UNet3d::forward: input and model are on different devices");
    auto x = inputTensor.to(torch::kFloat32).contiguous();
    unet_plan* plan = plan_for(x.size(2), x.size(3), x.size(4));
    auto ws = workspace_for(plan);
    if (is_training() && torch::GradMode::is_enabled())
        return UNetForwardFn::apply(x, trigger_, reinterpret_cast<int64_t>(this), reinterpret_cast<int64_t>(plan), ws);
    return run_forward(plan, ws, x, is_training() ? 1 : 0);
}

// ---- unet.cpp:7-22 ------------------------------------------------------------------------------------------
void UNet3dImpl::prepare_for_inference(const torch::Device& device)
{
    to_device(device);
    eval();
    for (size_t i = 0; i + 1 < buffers_.size(); i += 2) {
        buffers_[i].zero_();          // running_mean
        buffers_[i + 1].fill_(1.0f);  // running_var
    }
}

// ---- unet.cpp:195-222 ---------------------------------------------------------------------------------------
void UNet3dImpl::copy_from(const UNet3dImpl& r)
{
    torch::NoGradGuard no_grad;
    ensure_flat();
    const_cast<UNet3dImpl&>(r).ensure_flat();
    if (flat_params.sizes() == r.flat_params.sizes())
        flat_params.copy_(r.flat_params);
    else {
        auto rhs = r.parameters(), lhs = parameters();
        for (size_t i = 0; i < rhs.size() && i < lhs.size(); ++i)
            if (lhs[i].sizes() == rhs[i].sizes()) lhs[i].copy_(rhs[i]);
    }
    for (size_t i = 0; i < r.buffers_.size() && i < buffers_.size(); ++i)
        if (buffers_[i].sizes() == r.buffers_[i].sizes()) buffers_[i].copy_(r.buffers_[i]);
    voxel_size = r.voxel_size;
    dim = r.dim;
    fov_strategy = r.fov_strategy;
    postproc = r.postproc;
    preproc = r.preproc;
}

// ---- unet.cpp:224-244: in-process reduce-to-root (across processes: one RCCL all-reduce of flat_grads) ----------
void UNet3dImpl::add_gradient_from(const UNet3dImpl& r)
{
    torch::NoGradGuard no_grad;
    ensure_flat();
    rebind_grads();
    const_cast<UNet3dImpl&>(r).ensure_flat();
    const_cast<UNet3dImpl&>(r).rebind_grads();
    flat_grads.add_(r.flat_grads.to(flat_grads.device()).to(torch::kFloat32));
}

// ---- unet.cpp:246-277 ---------------------------------------------------------------------------------------
void UNet3dImpl::create_optimizer(float learning_rate)
{
    std::vector<torch::Tensor> decay_params, no_decay_params;
    for (auto& p : named_parameters()) {
        auto v = p.value();
        const auto& name = p.key();
        bool no_decay = name.find("bias") != std::string::npos || v.dim() <= 1;
        (no_decay ? no_decay_params : decay_params).push_back(v);
    }
    std::vector<torch::optim::OptimizerParamGroup> groups;
    auto opt_d = std::make_unique<torch::optim::SGDOptions>(learning_rate);
    opt_d->momentum(0.99); opt_d->nesterov(true); opt_d->weight_decay(3e-5);
    auto opt_nd = std::make_unique<torch::optim::SGDOptions>(learning_rate);
    opt_nd->momentum(0.99); opt_nd->nesterov(true); opt_nd->weight_decay(0.0);
    groups.push_back(torch::optim::OptimizerParamGroup(decay_params, std::move(opt_d)));
    groups.push_back(torch::optim::OptimizerParamGroup(no_decay_params, std::move(opt_nd)));
    optimizer = std::make_shared<torch::optim::SGD>(groups, torch::optim::SGDOptions(learning_rate));
}

// ---- unet.cpp:279-291 ---------------------------------------------------------------------------------------
std::string UNet3dImpl::get_info(void) const
{
    std::ostringstream out;
    out << "in: " << in_count << " out: " << out_count << std::endl;
    out << "dim: " << dim << " reso: " << voxel_size << std::endl;
    out << "structure: " << architecture << std::endl;
    if (!preproc.empty()) out << "preproc: " << preproc << std::endl;
    if (!postproc.empty()) out << "postproc: " << postproc << std::endl;
    return out.str();
}

// ---- unet.cpp:293-304 ---------------------------------------------------------------------------------------
void UNet3dImpl::print_layers(void)
{
    for (auto& p : named_parameters()) std::cout << p.key() << " " << p.value().sizes() << std::endl;
}

// ---- fused pieces: train.cpp:634-706 (losses + backward) and train.cpp:759-766 (step epilogue) -----------------
torch::Tensor UNet3dImpl::loss_and_backward(torch::Tensor input, torch::Tensor target, bool ce, bool dice, bool mse, int collapse_before)
{
    ensure_flat();
    auto x = input.to(torch::kFloat32).contiguous();
    unet_plan* plan = plan_for(x.size(2), x.size(3), x.size(4));
    auto ws = workspace_for(plan);
    auto outs = run_forward(plan, ws, x, 1);
    size_t sb = 0;
    unet_loss_scratch_bytes(plan, &sb);
    auto sc = torch::empty({(int64_t)sb}, torch::TensorOptions().dtype(torch::kUInt8).device(x.device()));
    auto losses = torch::empty({4}, torch::TensorOptions().dtype(torch::kFloat32).device(x.device()));
    std::vector<torch::Tensor> gouts;
    std::vector<const float*> op;
    std::vector<float*> gp;
    for (auto& o : outs) {
        gouts.push_back(o.defined() ? torch::empty_like(o) : torch::Tensor());
        op.push_back(o.defined() ? o.data_ptr<float>() : nullptr);
        gp.push_back(o.defined() ? gouts.back().data_ptr<float>() : nullptr);
    }
    auto t = target.to(torch::kLong).contiguous();
    int mask = (ce ? 1 : 0) | (dice ? 2 : 0) | (mse ? 4 : 0);
    check(unet_loss(plan, op.data(), t.data_ptr<int64_t>(), mask, collapse_before, gp.data(), losses.data_ptr<float>(), sc.data_ptr(),
                    stream_of(x.device())));
    run_backward(plan, ws, gouts);
    return losses;
}

// ---- data parallel over RCCL: the C++ host's replacement of train.cpp:573-579 (copy_from per step) and :756-757 (add_gradient_from) ----
void UNet3dImpl::broadcast_parameters(int root)
{
    if (!comm_) throw std::runtime_error("broadcast_parameters: no communicator attached");
    ensure_flat();
    void* st = stream_of(flat_params.device());
    check(unet_comm_broadcast(comm_, flat_params.data_ptr<float>(), flat_params.numel(), root, st));
    for (auto& b : buffers_) check(unet_comm_broadcast(comm_, b.data_ptr<float>(), b.numel(), root, st));   // unet.cpp:207-215
    check(unet_comm_join(comm_, st));
}

// BatchNorm running statistics: copy_from overwrites every replica's buffers with the root's each step (unet.cpp:207-215,
// train.cpp:573-579), so only the root's statistics exist in the reference; the same here (validation / a checkpoint agree on any rank)
void UNet3dImpl::broadcast_buffers(int root)
{
    if (!comm_) throw std::runtime_error("broadcast_buffers: no communicator attached");
    if (buffers_.empty()) return;
    ensure_flat();
    void* st = stream_of(flat_params.device());
    for (auto& b : buffers_) check(unet_comm_broadcast(comm_, b.data_ptr<float>(), b.numel(), root, st));
    check(unet_comm_join(comm_, st));
}

void UNet3dImpl::allreduce_gradients(void)
{
    if (!comm_) throw std::runtime_error("allreduce_gradients: no communicator attached");
    ensure_flat();
    rebind_grads();
    void* st = stream_of(flat_grads.device());
    const int64_t hi = reduced_from_ >= 0 ? reduced_from_ : flat_grads.numel();   // buckets above were started by the overlapped backward
    check(unet_allreduce_grads(comm_, flat_grads.data_ptr<float>(), 0, hi, st));
    // the forwards of this step have updated each rank's own running statistics: they follow rank 0, every step, like the Python trainer
    for (auto& b : buffers_) check(unet_comm_broadcast(comm_, b.data_ptr<float>(), b.numel(), 0, st));
    check(unet_comm_join(comm_, st));
    reduced_from_ = -1;
}

// loss_and_backward for a rank's LAST micro-step of an optimizer step: the backward runs in buckets (unet_plan_backward_buckets) and the
// all-reduce of each finished bucket starts at once on the communicator's stream; the last bucket is left to allreduce_gradients()
torch::Tensor UNet3dImpl::loss_and_backward_overlapped(torch::Tensor input, torch::Tensor target, bool ce, bool dice, bool mse, int collapse_before)
{
    if (!comm_) return loss_and_backward(input, target, ce, dice, mse, collapse_before);
    ensure_flat();
    rebind_grads();
    auto x = input.to(torch::kFloat32).contiguous();
    unet_plan* plan = plan_for(x.size(2), x.size(3), x.size(4));
    auto ws = workspace_for(plan);
    auto outs = run_forward(plan, ws, x, 1);
    size_t sb = 0;
    unet_loss_scratch_bytes(plan, &sb);
    auto sc = torch::empty({(int64_t)sb}, torch::TensorOptions().dtype(torch::kUInt8).device(x.device()));
    auto losses = torch::empty({4}, torch::TensorOptions().dtype(torch::kFloat32).device(x.device()));
    std::vector<torch::Tensor> gouts;
    std::vector<const float*> op, go;
    std::vector<float*> gp;
    for (auto& o : outs) {
        gouts.push_back(o.defined() ? torch::empty_like(o) : torch::Tensor());
        op.push_back(o.defined() ? o.data_ptr<float>() : nullptr);
        gp.push_back(o.defined() ? gouts.back().data_ptr<float>() : nullptr);
        go.push_back(gp.back());
    }
    auto t = target.to(torch::kLong).contiguous();
    int mask = (ce ? 1 : 0) | (dice ? 2 : 0) | (mse ? 4 : 0);
    void* st = stream_of(x.device());
    check(unet_loss(plan, op.data(), t.data_ptr<int64_t>(), mask, collapse_before, gp.data(), losses.data_ptr<float>(), sc.data_ptr(), st));
    std::vector<const float*> pp;
    std::vector<float*> gw;
    int64_t off = 0;
    for (auto& p : params_) { pp.push_back(p.data_ptr<float>()); gw.push_back(flat_grads.data_ptr<float>() + off); off += p.numel(); }
    int nb = 0, op_lo[8];
    int64_t elem_lo[8];
    check(unet_plan_backward_buckets(plan, 3, &nb, op_lo, elem_lo));
    int op_hi = 1 << 30;
    int64_t elem_hi = flat_grads.numel();
    for (int k = 0; k < nb; ++k) {
        check(unet_backward_part(plan, pp.data(), go.data(), gw.data(), nullptr, ws.data_ptr(), op_hi, op_lo[k], st));
        if (k + 1 < nb) {   // finished bucket: its sum over the ranks runs under the next part of the backward
            check(unet_allreduce_grads(comm_, flat_grads.data_ptr<float>(), elem_lo[k], elem_hi, st));
            reduced_from_ = elem_lo[k];
        }
        op_hi = op_lo[k]; elem_hi = elem_lo[k];
    }
    return losses;
}

// ---- train.cpp:787 (torch::save(*optimizer, model_path + ".opt")) and :945-957 (torch::load on resume) ------------------------------
// The fused update (sgd_step) keeps its momentum in ONE flat buffer.  So that the optimizer object the callers hold stays the
// resume path, the per-parameter `momentum_buffer` tensors of torch::optim::SGD's state are made aliases of that buffer: whichever
// of optimizer->step() (train.cpp:765) and sgd_step() runs, both read and write the same momentum, and torch::save / torch::load of
// *optimizer carry it.  After a torch::load the state holds fresh tensors: bind_optimizer_state() copies them in and re-aliases.
void UNet3dImpl::bind_optimizer_state(void)
{
    ensure_flat();
    if (!momentum_.defined() || momentum_.device() != flat_params.device() || momentum_.numel() != flat_params.numel())
        momentum_ = momentum_.defined() && momentum_.numel() == flat_params.numel() ? momentum_.to(flat_params.device()) : torch::zeros_like(flat_params);
    if (!optimizer) return;
    torch::NoGradGuard ng;
    auto& st = optimizer->state();
    auto opts = torch::TensorOptions().dtype(torch::kFloat32).device(momentum_.device());
    int64_t off = 0;
    for (auto& p : params_) {
        float* slot = momentum_.data_ptr<float>() + off;
        off += p.numel();
        void* key = p.unsafeGetTensorImpl();
        auto it = st.find(key);
        if (it == st.end()) {
            auto ps = std::make_unique<torch::optim::SGDParamState>();
            ps->momentum_buffer(torch::from_blob(slot, p.sizes(), opts));
            st[key] = std::move(ps);
            continue;
        }
        auto& ps = static_cast<torch::optim::SGDParamState&>(*it->second);
        if (ps.momentum_buffer().defined() && ps.momentum_buffer().data_ptr() == (void*)slot) continue;
        auto alias = torch::from_blob(slot, p.sizes(), opts);
        if (ps.momentum_buffer().defined()) alias.copy_(ps.momentum_buffer().reshape(p.sizes()));
        ps.momentum_buffer(alias);
    }
}

bool UNet3dImpl::save_optimizer(const std::string& file_name)
{
    try {
        if (!optimizer) throw std::runtime_error("save_optimizer: no optimizer (create_optimizer first)");
        bind_optimizer_state();
        torch::save(*optimizer, file_name);
        return true;
    } catch (const std::exception& e) { error_msg = std::string("cannot save optimizer: ") + e.what(); return false; }
}

bool UNet3dImpl::load_optimizer(const std::string& file_name)
{
    try {
        if (!optimizer) throw std::runtime_error("load_optimizer: no optimizer (create_optimizer first)");
        torch::load(*optimizer, file_name);
        bind_optimizer_state();
        return true;
    } catch (const std::exception& e) { error_msg = std::string("cannot load optimizer: ") + e.what(); return false; }   // train.cpp:953-955
}

void UNet3dImpl::sgd_step(float lr, float grad_scale, float clip_norm)
{
    ensure_flat();
    rebind_grads();
    bind_optimizer_state();
    if (!scratch_.defined()) scratch_ = torch::empty({65536 + 16}, torch::TensorOptions().dtype(torch::kUInt8).device(flat_params.device()));
    if (plans_.empty()) throw std::runtime_error("sgd_step before any forward");
    check(unet_sgd_step(plans_.begin()->second, flat_params.data_ptr<float>(), flat_grads.data_ptr<float>(), momentum_.data_ptr<float>(), lr,
                        0.99f, 1, 3e-5f, clip_norm, grad_scale, (float*)((char*)scratch_.data_ptr() + 65536), scratch_.data_ptr(),
                        stream_of(flat_params.device())));
}
