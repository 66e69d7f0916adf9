"""Generates tests/golden/*.npz with the ATen CPU kernels (oracle/aten_ref.py), i.e. with the third-party
dependency (libtorch) that holds the reference's arithmetic for this path, driven in unet.cpp's order.
The reference itself is not buildable in this image (needs TIPL) and holds no tests or
golden vectors for this path, so these fixtures + the live ATen comparison are what pins
the C oracle.  Run:  python tests/golden/make_golden.py   (CPU only, ~1 min)
"""
import os
import sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import aten_ref as A  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

# config 1 of BASELINE.json: 2-level UNet3d, 8 base channels (bnorm/relu/max_pool variant)
ARCH_BN = ("conv8,ks3,stride1+bnorm,relu+conv8,ks3,stride1+bnorm,relu\n"
           "max_pool+conv16,ks3,stride1+bnorm,relu+conv16,ks3,stride1+bnorm,relu+conv_trans8,ks2,stride2\n"
           "conv8,ks3,stride1+bnorm,relu+conv8,ks3,stride1+bnorm,relu+conv6,ks1,stride1")
# every layer kind of unet.cpp:24-101 once: stride-2 conv, norm, elu/leaky/relu, max_pool, upsample, conv_trans, 2 heads
ARCH_MIX = ("conv8,ks3,stride1+norm,elu+conv8,ks3,stride1+norm,leaky_relu\n"
            "conv16,ks3,stride2+norm,elu+conv16,ks3,stride1+norm,leaky_relu\n"
            "max_pool+conv16,ks3,stride1+norm,relu+upsample\n"
            "conv16,ks3,stride1+norm,leaky_relu+conv3,ks1,stride1+conv_trans8,ks2,stride2\n"
            "conv8,ks3,stride1+norm,leaky_relu+conv3,ks1,stride1")
# MFMA-eligible channel counts (multiples of 16) on a 3-level net
ARCH_16 = ("conv16,ks3,stride1+norm,leaky_relu+conv16,ks3,stride1+norm,leaky_relu\n"
           "conv32,ks3,stride2+norm,leaky_relu+conv32,ks3,stride1+norm,leaky_relu\n"
           "conv64,ks3,stride2+norm,leaky_relu+conv64,ks3,stride1+norm,leaky_relu+conv_trans32,ks2,stride2\n"
           "conv32,ks3,stride1+norm,leaky_relu+conv32,ks3,stride1+norm,leaky_relu+conv4,ks1,stride1+conv_trans16,ks2,stride2\n"
           "conv16,ks3,stride1+norm,leaky_relu+conv16,ks3,stride1+norm,leaky_relu+conv4,ks1,stride1")


def make_model(arch, cin, cout, seed=0, perturb=True):
    torch.manual_seed(seed)
    m = A.UNet3dRef(cin, cout, arch)
    if perturb:  # make norm gamma/beta non-trivial
        g = torch.Generator().manual_seed(seed + 1)
        with torch.no_grad():
            for p in m.parameters():
                if p.dim() == 1:
                    p.add_(0.1 * torch.randn(p.shape, generator=g))
    return m


def small_case(name, arch, cin, cout, n, lr=0.01, batch_size=1):
    m = make_model(arch, cin, cout)
    m.train()
    x, t = A.synthetic_sample(cin, cout, (n, n, n), 1)
    d = {"arch": np.array(arch), "cin": cin, "cout": cout, "x": x[0].numpy(), "target": t[0].numpy(), "lr": lr,
         "batch_size": batch_size}
    for i, p in enumerate(m.parameters()):
        d["param%d" % i] = p.detach().numpy().copy()
    outs = m(x)
    loss, stats = A.deep_supervision_loss(outs, t, cout)
    loss.backward()
    for k, o in enumerate(outs):
        d["logits%d" % k] = o[0].detach().numpy()
    d["loss"] = float(loss.detach())
    d["stats"] = np.array([float(s) for s in stats])
    for i, p in enumerate(m.parameters()):
        d["grad%d" % i] = p.grad.numpy().copy()
    for i, b in enumerate(m.buffers()):
        d["buffer_after%d" % i] = b.numpy().copy()
    opt = m.create_optimizer(lr)
    d["grad_norm"] = float(A.train_step_epilogue(m, opt, batch_size))
    for i, p in enumerate(m.parameters()):
        d["param_after%d" % i] = p.detach().numpy().copy()
    # eval-mode logits: (a) validation path train.cpp:834-840 (running stats as they are), (b) prepare_for_inference
    m.eval()
    with torch.no_grad():
        d["eval_logits0"] = m(x)[0][0].numpy()
        m.prepare_for_inference()
        d["infer_logits0"] = m(x)[0][0].numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    print(name, "loss", d["loss"], "stats", d["stats"], "gnorm", d["grad_norm"])


def logit_stride(n, level):
    """sampling stride of the stored logits of a level (full tensors are kept up to 16^3 voxels; the L2 norm covers the rest)"""
    side = n >> level
    st = 1
    while side // st > 16:
        st *= 2
    return st if n > 64 else (4 if level == 0 else 1)


GRAD_SAMPLE_STRIDE = 997


def default_case(n=64, variant=""):
    """default architecture (train.cpp:1054-1069), in=1, out=6, weights = module init under manual_seed(0)
    (as the GUI does, mainwindow_training.cpp:253).  Params are 60 MB, so the fixture holds a param checksum,
    sampled logits, per-tensor grad norms and the leading elements of every grad.
    variant "elu": every leaky_relu replaced by elu (a smooth activation: fp32 gradients agree to rounding, no kink voxels);
    variant "bf16": the same ATen kernels with bf16 roundings where the engine's bf16 configuration stores bf16
    (oracle/aten_ref.py:run_bf16_storage) -- the checker for the benchmarked configuration."""
    arch = A.default_feature(6)
    if variant == "elu":
        arch = arch.replace("leaky_relu", "elu")
    m = make_model(arch, 1, 6, perturb=False)
    m.bf16_storage = variant == "bf16"
    m.train()
    x, t = A.synthetic_sample(1, 6, (n, n, n), 1)
    outs = m(x)
    loss, stats = A.deep_supervision_loss(outs, t, 6)
    loss.backward()
    d = {"n": n, "loss": float(loss.detach()), "stats": np.array([float(s) for s in stats])}
    d["param_l2"] = np.array([float(p.detach().double().norm()) for p in m.parameters()])
    d["param_head"] = np.array([float(p.detach().flatten()[0]) for p in m.parameters()])
    for k, o in enumerate(outs):
        a = o[0].detach().numpy()
        st = logit_stride(n, k)
        d["logits%d" % k] = a[:, ::st, ::st, ::st].copy()
        d["logits_l2_%d" % k] = float(np.sqrt((a.astype(np.float64) ** 2).sum()))
    d["grad_l2"] = np.array([float(p.grad.double().norm()) for p in m.parameters()])
    d["grad_head"] = np.stack([np.pad(p.grad.flatten()[:16].numpy(), (0, max(0, 16 - p.numel()))) for p in m.parameters()])
    # a strided sample of EVERY parameter gradient (every GRAD_SAMPLE_STRIDE-th element of each flattened tensor, concatenated in
    # parameters() order) and each tensor's largest magnitude: a wrong-but-norm-preserving gradient (a permuted tap, a transposed
    # channel pair) changes these although it leaves grad_l2 and the 16 leading elements alone
    d["grad_sample"] = np.concatenate([p.grad.flatten()[::GRAD_SAMPLE_STRIDE].numpy() for p in m.parameters()])
    d["grad_absmax"] = np.array([float(p.grad.abs().max()) for p in m.parameters()])
    np.savez_compressed(os.path.join(HERE, "default_arch_%d%s.npz" % (n, "_" + variant if variant else "")), **d)
    print("default", n, variant, "loss", d["loss"], "stats", d["stats"])


if __name__ == "__main__":
    torch.set_num_threads(8)
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    if only == "variants":      # python tests/golden/make_golden.py variants: only the elu / bf16-storage fixtures of the default architecture
        default_case(64, "elu")
        default_case(64, "bf16")
        default_case(128, "bf16")
        sys.exit(0)
    small_case("cfg1_bnorm_16", ARCH_BN, 1, 6, 16)
    small_case("mix_16", ARCH_MIX, 2, 3, 16)
    small_case("c16_24", ARCH_16, 1, 4, 24, batch_size=2)
    default_case(64)
    default_case(128)   # BASELINE.json's size (config 2: the fp32 parity configuration); ~1 min, ~12 GB
    if only in ("", "variants"):
        default_case(64, "elu")
        default_case(64, "bf16")
        default_case(128, "bf16")   # the benchmarked configuration (BASELINE.json configs[2])
