"""CPU: the C-ABI library loads, exports every symbol include/unet_hip.h declares, and its host logic
(DSL parser, graph lowering, parameter order, shapes, FLOP counts) matches the oracle restatements.
No compute calls here (no GPU)."""
import ctypes
import math
import os
import re

import pytest

import unet_studio_amd as U
from oracle import aten_ref as A
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(U.engine.LIB_PATH)
    for header, exports in (("unet_hip.h", U.engine.EXPORTS), ("unet_augment.h", U.augment.EXPORTS)):
        hdr = open(os.path.join(ROOT, "include", header)).read()
        declared = set(re.findall(r"\b(unet_[a-z0-9_]+)\s*\(", hdr))
        declared -= {"unet_plan"}
        assert declared, "no declarations parsed"
        for name in sorted(declared):
            assert hasattr(lib, name), "libunet_hip.so does not export " + name
        assert declared == set(exports)


def test_plan_matches_reference_parameter_order_default_arch():
    arch = A.default_feature(6)
    p = U.Plan(arch, 1, 6, (128, 128, 128))
    ref = A.UNet3dRef(1, 6, arch)
    assert p.param_shapes == [tuple(q.shape) for q in ref.parameters()]
    assert len(p.param_shapes) == 108 and sum(math.prod(s) for s in p.param_shapes) == 15025822
    # weight decay mask, unet.cpp:254
    names = [n for n, _ in ref.named_parameters()]
    assert p.param_decay == [not ("bias" in n or q.dim() <= 1) for n, q in ref.named_parameters()], names
    assert p.param_names == names     # named_parameters() keys: "encode0.0.weight", ..., "output0.0.bias"
    assert p.output_shapes == [(1, 6, 128 >> l, 128 >> l, 128 >> l) for l in range(5)]
    # BASELINE.md §4 algorithmic work
    assert abs(p.flops_fwd - 245.165e9) < 0.01e9 and abs(p.flops_bwd - 488.518e9) < 0.01e9
    assert p.buffer_numel == []


@pytest.mark.parametrize("arch,cin,cout", [
    ("conv8,ks3,stride1+bnorm,relu+conv8,ks3,stride1+bnorm,relu\n"
     "max_pool+conv16,ks3,stride1+bnorm,relu+conv16,ks3,stride1+bnorm,relu+conv_trans8,ks2,stride2\n"
     "conv8,ks3,stride1+bnorm,relu+conv8,ks3,stride1+bnorm,relu+conv6,ks1,stride1", 1, 6),
    ("conv8,ks3,stride1+norm,elu+conv8,ks3,stride1+norm,leaky_relu\n"
     "conv16,ks3,stride2+norm,elu+conv16,ks3,stride1+norm,leaky_relu\n"
     "max_pool+conv16,ks3,stride1+norm,relu+upsample\n"
     "conv16,ks3,stride1+norm,leaky_relu+conv3,ks1,stride1+conv_trans8,ks2,stride2\n"
     "conv8,ks3,stride1+norm,leaky_relu+conv3,ks1,stride1", 2, 3),
    # head token with an activation and tokens that force materialisation (act after act, norm on a skip)
    ("conv4,ks3,stride1,relu+norm,relu\nnorm+conv8,ks3,stride2,elu\nmax_pool+conv4,ks3,stride1+upsample\n"
     "conv4,ks3,stride1+conv2,ks1,stride1,relu+conv_trans4\nconv4+conv2,ks1,stride1,relu", 1, 2),
])
def test_plan_shapes_match_oracles(arch, cin, cout):
    try:
        ref = A.UNet3dRef(cin, cout, arch)
    except RuntimeError:
        ref = None
    o = O.OracleUNet(cin, cout, arch)
    p = U.Plan(arch, cin, cout, (16, 16, 16), U.DTYPE_F32)
    assert p.param_shapes == [q.shape for q in o.params]
    if ref is not None:
        assert p.param_shapes == [tuple(q.shape) for q in ref.parameters()]
    assert len(p.buffer_numel) == len([b for b in o.buffers if b.dtype.kind == "f"])
    assert p.workspace_bytes > 0 and "conv" in p.describe()


def test_dsl_errors_match_reference_messages():
    for arch, msg in (("conv8\nconv8", "invalid u-net structure"),
                      ("conv8,ks5\nconv8\nconv8", "conv supports only ks1 stride1, ks3 stride1, and ks3 stride2"),
                      ("conv8\nconv_trans8,ks3\nconv8", "conv_trans supports only ks2 stride2"),
                      ("conv8\nfoo7\nconv8", "unknown layer: foo"),
                      ("conv8\nconv8\nconv8\nconv8", "invalid u-net structure")):
        with pytest.raises(U.UNetError, match=msg):
            U.Plan(arch, 1, 2, (16, 16, 16))
    with pytest.raises(U.UNetError, match="size mismatch"):  # skip and decoder input differ in size
        U.Plan("conv4\nconv4,ks3,stride2\nconv4", 1, 2, (16, 16, 16))
    with pytest.raises(U.UNetError, match="channel mismatch"):
        U.Plan("conv4\nconv4\nconv2,ks1+conv2,ks1", 1, 2, (8, 8, 8))


def test_crlf_and_blank_lines():
    a = "conv4\r\n\r\nconv4\r\nconv2,ks1\r\n"
    assert U.Plan(a, 1, 2, (8, 8, 8)).param_shapes == U.Plan("conv4\nconv4\nconv2,ks1", 1, 2, (8, 8, 8)).param_shapes


def test_lowered_graph_fuses_norm_and_activation():
    d = U.Plan(A.default_feature(6), 1, 6, (64, 64, 64)).describe()
    assert "materialize" not in d          # every norm/act of the default architecture is read-fused
    assert d.count("=> results[") == 5     # 5 heads write forward()'s results directly
    assert "src[t2+norm+leaky_relu,t29]" in d  # cat(skip, x) is a dual-source read, never materialised


def test_bench_launches_its_own_ranks_and_names_the_missing_devices():
    """bench.py --gpus N without WORLD_SIZE is its own launcher (train.cpp:581-606 starts one worker per device): with fewer
    than N devices it must stop with a message that says so -- not ask for torch.distributed.run -- and a non-zero exit code."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "64 devices needed" in (r.stderr + r.stdout)
    assert "torch.distributed.run" not in (r.stderr + r.stdout)


def test_plan_op_list_names_every_conv_of_the_default_arch():
    p = U.Plan(A.default_feature(6), 1, 6, (128, 128, 128))
    ops = p.ops()
    convs = [o for o in ops if o["kind"] in (1, 2)]
    assert len(convs) == 22 + 5 + 5          # 22 3x3x3 convs, 5 heads, 5 conv_trans
    fl = 0.0
    for o in convs:
        v = o["out_dims"] if o["kind"] == 1 else o["in_dims"]
        fl += 2.0 * (o["ks"] ** 3 if o["kind"] == 1 else 8) * o["cin"] * o["cout"] * v[0] * v[1] * v[2]
    assert abs(fl - p.flops_fwd) < 1e-6 * p.flops_fwd


def test_hand_counted_kernels_passed_the_build_time_assembly_check():
    """k_mfma_conv_z, k_mfma_wgrad_z and the LDS-DMA kernels k_mfma_conv_z16 / _z32 wait for inline-assembly loads with hand-counted
    vmcnt values; csrc/build.sh runs csrc/tools/check_asm_loads.py on their device assembly and records the result next to the
    library.  The library that is loaded must be one whose checks all passed."""
    import glob
    import json
    recs = sorted(glob.glob(os.path.join(ROOT, "unet-studio_amd", "asm_loads_check_*.json")))
    assert {os.path.basename(r) for r in recs} >= {"asm_loads_check_conv_z.json", "asm_loads_check_wgrad_z.json", "asm_loads_check_conv_zdma.json"}
    for r in recs:
        d = json.load(open(r))
        assert d["ok"] and d["kernels"] and all(k["ok"] for k in d["kernels"]), r
    dma = json.load(open(os.path.join(ROOT, "unet-studio_amd", "asm_loads_check_conv_zdma.json")))
    assert {k["planes_ahead"] for k in dma["kernels"]} == {5}


def test_unet_hpp_keeps_the_reference_class_surface():
    """include/unet.hpp is a drop-in for the reference's unet.hpp: every member the reference's callers use (SURVEY.md section 8b;
    declarations of /root/reference/unet.hpp:13-70, normalised for white space) must still be declared with the same signature.  A
    static check of the header's text -- compiling against the reference's callers needs TIPL and Qt, which this image does not have."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = re.sub(r"\s+", " ", open(os.path.join(root, "include", "unet.hpp")).read())
    required = [
        "struct UNet3dImpl : torch::nn::Module",
        "int in_count = 1;", "int out_count = 1;",
        "std::string architecture,preproc,postproc,orientation,fov_strategy,error_msg;",
        "std::shared_ptr<torch::optim::SGD> optimizer;",
        "std::vector<float> testing_errors,training_errors;",
        "std::vector<unsigned int> single_component_label;",
        "mutable std::mutex error_mutex;",
        "auto get_training_errors(void) const", "auto get_testing_errors(void) const",
        "voxel_size = {1.0f,1.0f,1.0f};", "dim = {192,224,192};",
        "std::deque<torch::nn::Sequential> encoding,decoding,decoding_tail;",
        "std::vector<torch::nn::Sequential> output;",
        "int create_layer(torch::nn::Sequential& layers,const std::string& def, int in_c);",
        "std::string get_info(void) const;",
        "UNet3dImpl(void){}", "UNet3dImpl(int32_t in_count_,int32_t out_count_,std::string);",
        "void copy_from(const UNet3dImpl& r);", "void add_gradient_from(const UNet3dImpl& r);",
        "void create_optimizer(float learning_rate);",
        "std::vector<torch::Tensor> forward(torch::Tensor inputTensor);",
        "void set_requires_grad(bool req)", "virtual void train(bool on = true) override",
        "void print_layers(void);", "torch::Device device(void) const",
        "void prepare_for_inference(const torch::Device& device);",
        "TORCH_MODULE_IMPL(UNet3d, UNet3dImpl);",
    ]
    missing = [r for r in required if re.sub(r"\s+", " ", r) not in text]
    assert not missing, "include/unet.hpp no longer declares: %s" % missing
