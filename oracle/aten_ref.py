"""ORACLE (test infrastructure, never shipped, never on the product path).

ATen-CPU restatement of the reference hot path.  The reference's arithmetic for
this path lives in a third-party dependency, libtorch (pins: 2.0.0+cu117 in
.github/workflows/build_unetstudio.yml:48, 1.13.0 in build_packages/docker/Dockerfile:18);
`unet.cpp` only builds a graph of `torch::nn` modules.  This file drives the SAME
ATen CPU kernels (torch 2.10.0 wheel of this image) in the SAME order:

  * `parse_token`   <- UNet3dImpl::create_layer         unet.cpp:24-101
  * `UNet3dRef`     <- UNet3dImpl::UNet3dImpl           unet.cpp:103-166
  * `forward`       <- UNet3dImpl::forward              unet.cpp:168-193
  * `prepare_for_inference`                             unet.cpp:7-22
  * `calc_losses`   <- calc_losses                      train.cpp:501-552
  * `deep_supervision_loss` <- train loop               train.cpp:634-706
  * `create_optimizer` / `train_step_epilogue`          unet.cpp:246-277, train.cpp:756-766

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
The reference itself cannot be built here: unet.hpp:10 includes TIPL, which is
neither vendored nor present, and writing a stand-in header is not allowed.
"""
import math
import torch
import torch.nn as nn
import torch.nn.functional as F


def split_lines(arch):
    # tipl::split_by_line_breaks (TIPL, not in tree): strip '\r', skip blank lines.
    return [l.strip("\r") for l in arch.split("\n") if l.strip("\r") != ""]


def parse_token(token):
    """unet.cpp:26-34 -- 'name<digits>' pairs separated by ','."""
    params = {}
    for arg in token.split(","):
        pos = next((i for i, ch in enumerate(arg) if ch.isdigit()), None)
        if pos is not None:
            params[arg[:pos]] = arg[pos:]
        else:
            params[arg] = "1"
    return params


class BatchNorm3dEps0(nn.BatchNorm3d):
    """torch::nn::BatchNorm3d(affine, track_running_stats, eps=0.0) of unet.cpp:80-84.  torch 2.10's Python
    wrapper F.batch_norm refuses eps == 0 (the C++ module the reference uses does not), so call the ATen
    op the C++ module calls, with the same bookkeeping (num_batches_tracked, momentum 0.1)."""

    def __init__(self, c):
        super().__init__(c, affine=True, track_running_stats=True, eps=1.0)
        self.eps = 0.0

    def forward(self, x):
        if self.training:
            self.num_batches_tracked.add_(1)
        return torch.batch_norm(x, self.weight, self.bias, self.running_mean, self.running_var,
                                self.training, self.momentum, 0.0, False)


def create_layer(layers, token, in_c):
    """unet.cpp:24-101. Appends modules to `layers` (list), returns out channels."""
    p = parse_token(token)
    out_c = in_c
    if "max_pool" in p:
        layers.append(nn.MaxPool3d(2, stride=2))
    elif "upsample" in p:
        layers.append(nn.Upsample(scale_factor=(2.0, 2.0, 2.0), mode="nearest"))
    elif "conv_trans" in p:
        out_c = int(p["conv_trans"])
        ks = int(p["ks"]) if "ks" in p else 2
        stride = int(p["stride"]) if "stride" in p else 2
        if ks != 2 or stride != 2:
            raise RuntimeError("conv_trans supports only ks2 stride2")
        layers.append(nn.ConvTranspose3d(in_c, out_c, ks, stride=stride))
    elif "conv" in p:
        out_c = int(p["conv"])
        ks = int(p["ks"]) if "ks" in p else 3
        stride = int(p["stride"]) if "stride" in p else 1
        if not ((ks == 1 and stride == 1) or (ks == 3 and stride in (1, 2))):
            raise RuntimeError("conv supports only ks1 stride1, ks3 stride1, and ks3 stride2")
        layers.append(nn.Conv3d(in_c, out_c, ks, stride=stride, padding=(ks - 1) // 2))
    elif "norm" in p:
        layers.append(nn.InstanceNorm3d(in_c, affine=True))
    elif "bnorm" in p:
        layers.append(BatchNorm3dEps0(in_c))
    else:
        raise RuntimeError("unknown layer: " + (token if not p else next(iter(p))))
    if "relu" in p:
        layers.append(nn.ReLU(inplace=True))
    elif "leaky_relu" in p:
        layers.append(nn.LeakyReLU(0.01, inplace=True))
    elif "elu" in p:
        layers.append(nn.ELU(inplace=True))
    return out_c


class _RoundBF16(torch.autograd.Function):
    """a tensor the MI355X engine STORES as bf16 (include/unet_hip.h, UNET_DTYPE_BF16): the value is rounded where it is stored, and
    so is its gradient (the gradient buffer of the same tensor is bf16 too)"""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


class _RoundFwdOnly(torch.autograd.Function):
    """a filter: the engine's matrix-core kernels read a bf16 pack of the fp32 master weights; the weight gradient is fp32"""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundGradOnly(torch.autograd.Function):
    """a view that is never stored (the engine's fused heads transform the raw tensor as they load it) but whose gradient is"""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


_ACTS = (nn.ReLU, nn.LeakyReLU, nn.ELU)


def run_bf16_storage(seq, x, last_view_unstored=False):
    """One nn.Sequential of unet.cpp:103-166 with the roundings of the engine's bf16 configuration (data layout): conv /
    conv_trans read bf16 filters and store their raw output as bf16; `norm + activation` is evaluated in fp32 on the stored raw tensor
    and its result (the activated copy every consumer reads) is stored as bf16; gradients are rounded where those tensors' gradient
    buffers are written.  Arithmetic between the roundings is ATen's fp32.  last_view_unstored: the sequence's last view is read by a
    fused head only (the decoder's full-resolution output), which keeps no copy."""
    mods = list(seq)
    views = [i for i, m in enumerate(mods) if isinstance(m, (nn.InstanceNorm3d, nn.BatchNorm3d) + _ACTS)]
    last_view = views[-1] if views else -1
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.ConvTranspose3d):
            x = _RoundBF16.apply(F.conv_transpose3d(x, _RoundFwdOnly.apply(m.weight), m.bias, m.stride))
        elif isinstance(m, nn.Conv3d):
            x = _RoundBF16.apply(F.conv3d(x, _RoundFwdOnly.apply(m.weight), m.bias, m.stride, m.padding))
        elif isinstance(m, (nn.InstanceNorm3d, nn.BatchNorm3d) + _ACTS):
            x = m(x)
            if not isinstance(m, _ACTS) and i + 1 < len(mods) and isinstance(mods[i + 1], _ACTS):
                i += 1
                x = mods[i](x)
            x = _RoundGradOnly.apply(x) if (last_view_unstored and i >= last_view) else _RoundBF16.apply(x)
        else:
            x = m(x)
        i += 1
    return x


class UNet3dRef(nn.Module):
    """unet.cpp:103-166; module registration order fixes parameters() order.
    bf16_storage = True (test infrastructure for the engine's bf16 configuration, not a reference behaviour): forward() rounds to bf16
    exactly where the engine stores bf16 (run_bf16_storage); the reference itself is fp32 throughout."""
    bf16_storage = False

    def __init__(self, in_count, out_count, architecture):
        super().__init__()
        self.in_count, self.out_count, self.architecture = in_count, out_count, architecture
        lines = split_lines(architecture)
        if len(lines) < 3:
            raise RuntimeError("invalid u-net structure")
        enc_count = len(lines) // 2 + 1
        enc_tokens = [l.split("+") for l in lines[:enc_count]]
        dec_tokens = [l.split("+") for l in lines[enc_count:]]
        if len(dec_tokens) != enc_count - 1:
            raise RuntimeError("invalid u-net structure: needs an odd number of lines")
        self.encoding = []
        channel = in_count
        skip_channels = []
        for level, toks in enumerate(enc_tokens):
            layers = []
            for t in toks:
                channel = create_layer(layers, t, channel)
            seq = nn.Sequential(*layers)
            self.add_module("encode%d" % level, seq)
            self.encoding.append(seq)
            skip_channels.append(channel)
        nd = len(dec_tokens)
        self.decoding, self.output, self.decoding_tail = [None] * nd, [None] * nd, [None] * nd
        out_token = dec_tokens[-1][-1]
        for level in range(nd - 1, -1, -1):
            toks = dec_tokens[nd - 1 - level]
            after_out = False
            channel += skip_channels[level]
            dec, outl, tail = [], [], []
            for t in toks:
                if t == out_token:
                    create_layer(outl, t, channel)
                    after_out = True
                    continue
                channel = create_layer(tail if after_out else dec, t, channel)
            self.decoding[level] = nn.Sequential(*dec)
            self.output[level] = nn.Sequential(*outl)
            self.decoding_tail[level] = nn.Sequential(*tail)
            self.add_module("decode%d" % level, self.decoding[level])
            if len(outl):
                self.add_module("output%d" % level, self.output[level])
            if len(tail):
                self.add_module("decode_tail%d" % level, self.decoding_tail[level])

    def forward(self, x):
        """unet.cpp:168-193."""
        skips = []
        results = [None] * len(self.output)
        n = len(self.encoding)
        bf = self.bf16_storage
        if bf:
            x = x.to(torch.bfloat16).to(x.dtype)      # the engine's input pack
        for level in range(n):
            x = run_bf16_storage(self.encoding[level], x) if bf else self.encoding[level](x)
            if level < n - 1:
                skips.append(x)
        for level in range(n - 2, -1, -1):
            # a tensor with two readers: the reader that comes LATER in the forward writes the gradient buffer first (bf16), the earlier
            # one adds to it and rounds again -- so the later reader's contribution is rounded on its own
            x = torch.cat([_RoundGradOnly.apply(skips[level]) if bf else skips[level], x], 1)
            skips[level] = None
            if bf:   # heads (fp32 weights, fp32 results) read the view; only a head reads the last level's
                x = run_bf16_storage(self.decoding[level], x, last_view_unstored=len(self.output[level]) > 0 and not len(self.decoding_tail[level]))
            else:
                x = self.decoding[level](x)
            if len(self.output[level]):
                results[level] = self.output[level](x)
            if len(self.decoding_tail[level]):
                x = run_bf16_storage(self.decoding_tail[level], _RoundGradOnly.apply(x) if len(self.output[level]) else x) if bf else self.decoding_tail[level](x)
        return results

    def train(self, on=True):
        """unet.hpp:53-62: train() also flips requires_grad."""
        for p in self.parameters():
            p.requires_grad_(on)
        return super().train(on)

    def prepare_for_inference(self):
        """unet.cpp:7-22."""
        self.eval()
        for m in self.modules():
            if isinstance(m, nn.BatchNorm3d):
                m.running_mean.zero_()
                m.running_var.fill_(1.0)
                if m.num_batches_tracked is not None:
                    m.num_batches_tracked.zero_()

    def create_optimizer(self, lr):
        """unet.cpp:246-277."""
        decay, no_decay = [], []
        for name, v in self.named_parameters():
            (no_decay if ("bias" in name or v.dim() <= 1) else decay).append(v)
        return torch.optim.SGD(
            [dict(params=decay, weight_decay=3e-5), dict(params=no_decay, weight_decay=0.0)],
            lr=lr, momentum=0.99, nesterov=True)


def default_feature(out_count):
    """train.cpp:1054-1069 (a data string: the reference's default architecture)."""
    out = "conv%d,ks1,stride1" % out_count
    nl = "norm,leaky_relu"
    enc = []
    prev = None
    for i, c in enumerate([16, 32, 64, 128, 256, 256]):
        s = 1 if i == 0 else 2
        enc.append("conv%d,ks3,stride%d+%s+conv%d,ks3,stride1+%s" % (c, s, nl, c, nl))
    enc[-1] += "+conv_trans256,ks2,stride2"
    dec = []
    for c, up in [(256, 128), (128, 64), (64, 32), (32, 16)]:
        dec.append("conv%d,ks3,stride1+%s+conv%d,ks3,stride1+%s+%s+conv_trans%d,ks2,stride2" % (c, nl, c, nl, out, up))
    dec.append("conv16,ks3,stride1+%s+conv16,ks3,stride1+%s+%s" % (nl, nl, out))
    return "\n".join(enc + dec)


def calc_losses(pred_raw, target_indices, C, collapse_before=0):
    """train.cpp:501-552."""
    if collapse_before < 0 or collapse_before >= C:
        raise RuntimeError("invalid collapse_before")
    logits, target, out_C = pred_raw, target_indices, C
    if collapse_before:
        logits = torch.cat([torch.logsumexp(pred_raw[:, :collapse_before], 1, True),
                            pred_raw[:, collapse_before:C]], 1)
        target = torch.clamp_min(target_indices - collapse_before + 1, 0)
        out_C = C - collapse_before + 1
    valid = target_indices < C
    v = valid.to(logits.dtype)
    n = torch.clamp_min(v.sum(), 1.0)
    target = torch.where(valid, target, torch.zeros_like(target))
    ce = F.cross_entropy(logits, target, reduction="none")
    ce = (ce * v).sum() / n
    prob = torch.clamp(torch.softmax(logits, 1), 1e-6, 1.0 - 1e-6)
    target_prob = prob.gather(1, target.unsqueeze(1)).squeeze(1)
    mse = ((torch.sum(prob * prob, 1) - 2.0 * target_prob + 1.0) * v).sum() / n
    eps = torch.tensor(1e-5, dtype=pred_raw.dtype)
    dice_sum = torch.zeros((), dtype=pred_raw.dtype)
    for c in range(1, out_C):
        p = prob.select(1, c) * v
        m = (target == c).to(p.dtype) * v
        inter = torch.sum(p * m, (1, 2, 3))
        card = torch.sum(p + m, (1, 2, 3))
        dice_sum = dice_sum + torch.sum((2.0 * inter + eps) / (card + eps))
    dice = 1.0 - dice_sum / float(target.size(0) * max(1, out_C - 1))
    return ce, dice, mse


def downsample_target(active_target):
    """train.cpp:645-662: nearest interpolate through a float round trip."""
    d, h, w = (active_target.size(1) >> 1, active_target.size(2) >> 1, active_target.size(3) >> 1)
    if d <= 0 or h <= 0 or w <= 0:
        raise RuntimeError("deep supervision target size became zero")
    t = active_target.unsqueeze(1).to(torch.float32)
    return F.interpolate(t, size=(d, h, w), mode="nearest").squeeze(1).to(torch.long)


def deep_supervision_loss(outputs, target, out_count, cost_ce=True, cost_dice=True, cost_mse=True,
                          collapse_before=0):
    """train.cpp:634-706. Returns (total_loss, (ce,dice,mse) of level 0)."""
    weight_sum = sum(1.0 / (1 << k) for k in range(len(outputs)))
    inv = 1.0 / weight_sum
    total, stats = None, None
    active = target
    for k, o in enumerate(outputs):
        if k > 0:
            active = downsample_target(active)
        if o is None:
            raise RuntimeError("undefined deep supervision output at level %d" % k)
        if o.size(1) != out_count:
            raise RuntimeError("output channel mismatch at level %d" % k)
        ce, dice, mse = calc_losses(o, active, out_count, collapse_before)
        if k == 0:
            stats = (ce.detach(), dice.detach(), mse.detach())
        level = None
        for on, l in ((cost_ce, ce), (cost_dice, dice), (cost_mse, mse)):
            if on:
                level = l if level is None else level + l
        if level is None:
            level = ce
        level = level * ((1.0 / (1 << k)) * inv)
        total = level if total is None else total + level
    return total, stats


def train_step_epilogue(model, optimizer, batch_size, lr=None):
    """train.cpp:566-571,759-766: lr set, grad/batch, clip 12.0, SGD step, zero_grad."""
    if lr is not None:
        for g in optimizer.param_groups:
            g["lr"] = lr
    for p in model.parameters():
        if p.grad is not None:
            p.grad.div_(batch_size)
    gn = torch.nn.utils.clip_grad_norm_(list(model.parameters()), 12.0)
    optimizer.step()
    optimizer.zero_grad(set_to_none=False)
    return gn


def poly_lr(lr0, epoch, total):
    """train.cpp:566."""
    return lr0 * math.pow(1.0 - float(epoch) / total, 0.9)


def synthetic_sample(in_count, out_count, size, seed):
    """SURVEY §8d synthetic inputs: image U[0,1) fp32, labels U{0..out-1} int64."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand((1, in_count) + tuple(size), generator=g, dtype=torch.float32)
    t = torch.randint(0, out_count, (1,) + tuple(size), generator=g, dtype=torch.int64)
    return x, t
