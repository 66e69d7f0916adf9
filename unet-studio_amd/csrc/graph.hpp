// Lowered graph of one UNet3d: the architecture DSL of UNet3dImpl (unet.cpp:24-166) walked in the
// order of UNet3dImpl::forward (unet.cpp:168-193) and turned into a flat op list over channels-last
// tensors.  Host-only, no HIP.
//
// Fusion model: a tensor is stored RAW (what its producing kernel wrote).  A norm and/or an activation
// that follows it in the DSL is not executed as a pass of its own; it is recorded on the tensor
// (Tensor::norm, Tensor::act) and every consumer applies `act(x*scale[c]+shift[c])` while it reads.
// A norm/act that cannot be recorded (tensor already has an activation, or is already visible to
// another consumer) is preceded by an OP_MATERIALIZE that writes the transformed tensor out.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace unet {

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_LEAKY = 2, ACT_ELU = 3 };
enum OpKind { OP_PACK_INPUT, OP_CONV, OP_CONVT, OP_NORM, OP_MATERIALIZE, OP_MAXPOOL, OP_UPSAMPLE, OP_EXPORT };

struct Tensor {
    int C = 0, D = 0, H = 0, W = 0;
    int norm = -1;        // index into Graph::norms: consumers apply its scale/shift
    int act = ACT_NONE;   // then this activation
    bool frozen = false;  // visible to more than one consumer: no further norm/act may be recorded
    bool needs_grad = true;
    int64_t voxels() const { return (int64_t)D * H * W; }
    int64_t numel() const { return voxels() * C; }
};

struct Norm {
    int tensor = -1;
    int C = 0;
    bool batch = false;   // BatchNorm3d(eps 0) vs InstanceNorm3d(eps 1e-5)
    int gamma = -1, beta = -1;  // parameter indices
    int buffer = -1;      // index of running_mean in buffers (running_var = +1), bnorm only
    double eps = 1e-5;
};

struct Op {
    OpKind kind;
    int nsrc = 0;
    int src[2] = {-1, -1};  // tensor ids; two sources = channel concat {skip, x} (unet.cpp:181)
    int dst = -1;           // tensor id (or -1 when the conv writes an external output)
    int weight = -1, bias = -1;  // parameter indices
    int cin = 0, cout = 0, ks = 0, stride = 0;
    int norm = -1;          // OP_NORM: which norm; OP_CONV/...: norm whose statistics the op should emit
    int out_level = -1;     // >= 0: result is forward()'s results[level], fp32 NCDHW
    std::string name;
};

struct Param {
    std::vector<int64_t> shape;
    bool decay = false;       // unet.cpp:254
    int64_t fan_in = 0;       // conv / conv_trans default init bound
    bool norm_weight = false; // gamma (init 1)
    std::string name;
};

struct Graph {
    int in_c = 0, out_c = 0, D = 0, H = 0, W = 0;
    std::vector<Tensor> tensors;
    std::vector<Norm> norms;
    std::vector<Op> ops;
    std::vector<Param> params;
    std::vector<int64_t> buffers;  // numel of each fp32 buffer (running_mean, running_var per bnorm)
    struct Out { int C = 0, D = 0, H = 0, W = 0; int tensor = -1; };
    std::vector<Out> outputs;      // per decoder level
    double flops_fwd = 0, flops_bwd = 0;

    // throws std::runtime_error with the reference's messages (unet.cpp:53,66,88,117)
    static Graph build(const std::string& arch, int in_c, int out_c, int D, int H, int W);
    std::string describe() const;
};

}  // namespace unet
