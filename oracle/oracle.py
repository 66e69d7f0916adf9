"""ORACLE (test infrastructure, never shipped, never on the product path).

numpy + plain-C (oracle/unet_oracle.c) restatement of the reference hot path: the graph walk of
unet.cpp:103-193 with a hand-written reverse pass, the losses of train.cpp:501-552, the
deep-supervision loop of train.cpp:634-706 and the step epilogue of train.cpp:756-766.
Pinned in tests/test_oracle.py against the ATen CPU kernels (oracle/aten_ref.py) and tests/golden/.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

from .aten_ref import parse_token, split_lines

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """gcc-compiles the C restatement next to its source."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "unet_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-o", so, src, "-lm"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_grad_norm.restype = C.c_double
    return _LIB


def _f(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _fn(a):
    return None if a is None else _f(a)


ACT = {"relu": 1, "leaky_relu": 2, "elu": 3}


class Layer:
    def __init__(self, kind, **kw):
        self.kind = kind
        self.__dict__.update(kw)


def create_layer(layers, token, in_c):
    """unet.cpp:24-101 -> list of Layer; returns out channels."""
    p = parse_token(token)
    out_c = in_c
    if "max_pool" in p:
        layers.append(Layer("max_pool", C=in_c))
    elif "upsample" in p:
        layers.append(Layer("upsample", C=in_c))
    elif "conv_trans" in p:
        out_c = int(p["conv_trans"])
        ks = int(p.get("ks", 2)); stride = int(p.get("stride", 2))
        if ks != 2 or stride != 2:
            raise RuntimeError("conv_trans supports only ks2 stride2")
        layers.append(Layer("conv_trans", cin=in_c, cout=out_c, nparam=2))
    elif "conv" in p:
        out_c = int(p["conv"])
        ks = int(p.get("ks", 3)); stride = int(p.get("stride", 1))
        if not ((ks == 1 and stride == 1) or (ks == 3 and stride in (1, 2))):
            raise RuntimeError("conv supports only ks1 stride1, ks3 stride1, and ks3 stride2")
        layers.append(Layer("conv", cin=in_c, cout=out_c, ks=ks, stride=stride, nparam=2))
    elif "norm" in p:
        layers.append(Layer("norm", C=in_c, nparam=2))
    elif "bnorm" in p:
        layers.append(Layer("bnorm", C=in_c, nparam=2))
    else:
        raise RuntimeError("unknown layer: " + (token if not p else next(iter(p))))
    for a in ("relu", "leaky_relu", "elu"):
        if a in p:
            layers.append(Layer("act", act=ACT[a]))
            break
    return out_c


class OracleUNet:
    """unet.cpp:103-166.  self.seqs = sequences in module registration order; self.params / self.buffers
    are float32 numpy arrays in parameters() / buffers() order."""

    def __init__(self, in_count, out_count, architecture):
        self.in_count, self.out_count = in_count, out_count
        lines = split_lines(architecture)
        if len(lines) < 3:
            raise RuntimeError("invalid u-net structure")
        ne = len(lines) // 2 + 1
        enc_tok = [l.split("+") for l in lines[:ne]]
        dec_tok = [l.split("+") for l in lines[ne:]]
        if len(dec_tok) != ne - 1:
            raise RuntimeError("invalid u-net structure: needs an odd number of lines")
        self.encoding, order = [], []
        ch = in_count
        skip = []
        for toks in enc_tok:
            L = []
            for t in toks:
                ch = create_layer(L, t, ch)
            self.encoding.append(L); order.append(L); skip.append(ch)
        nd = len(dec_tok)
        self.decoding, self.output, self.tail = [None] * nd, [None] * nd, [None] * nd
        out_token = dec_tok[-1][-1]
        for level in range(nd - 1, -1, -1):
            toks = dec_tok[nd - 1 - level]
            after = False
            ch += skip[level]
            d, o, tl = [], [], []
            for t in toks:
                if t == out_token:
                    create_layer(o, t, ch); after = True
                    continue
                ch = create_layer(tl if after else d, t, ch)
            self.decoding[level], self.output[level], self.tail[level] = d, o, tl
            order += [d, o, tl]
        self.params, self.buffers, self.training = [], [], True
        for L in order:
            for ly in L:
                if ly.kind == "conv":
                    ly.pi = len(self.params)
                    self.params += [np.zeros((ly.cout, ly.cin) + (ly.ks,) * 3, np.float32), np.zeros(ly.cout, np.float32)]
                elif ly.kind == "conv_trans":
                    ly.pi = len(self.params)
                    self.params += [np.zeros((ly.cin, ly.cout, 2, 2, 2), np.float32), np.zeros(ly.cout, np.float32)]
                elif ly.kind in ("norm", "bnorm"):
                    ly.pi = len(self.params)
                    self.params += [np.ones(ly.C, np.float32), np.zeros(ly.C, np.float32)]
                    if ly.kind == "bnorm":
                        ly.bi = len(self.buffers)
                        self.buffers += [np.zeros(ly.C, np.float32), np.ones(ly.C, np.float32), np.zeros(1, np.int64)]

    def load_params(self, arrays, buffers=None):
        assert len(arrays) == len(self.params)
        for i, a in enumerate(arrays):
            a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
            assert a.shape == self.params[i].shape, (i, a.shape, self.params[i].shape)
            self.params[i] = a.copy()
        if buffers is not None:
            for i, b in enumerate(buffers):
                self.buffers[i] = np.array(b, copy=True).reshape(self.buffers[i].shape).astype(self.buffers[i].dtype)

    def prepare_for_inference(self):
        """unet.cpp:7-22."""
        self.training = False
        for i in range(0, len(self.buffers), 3):
            self.buffers[i][:] = 0; self.buffers[i + 1][:] = 1; self.buffers[i + 2][:] = 0

    # ---- one Sequential, forward with tape ----
    def _seq_fwd(self, L, x, tape):
        l = lib()
        for ly in L:
            Cc, D, H, W = x.shape
            if ly.kind == "conv":
                pad = (ly.ks - 1) // 2
                od = [(s + 2 * pad - ly.ks) // ly.stride + 1 for s in (D, H, W)]
                y = np.empty((ly.cout, *od), np.float32)
                l.orc_conv3d_fwd(_f(x), _f(self.params[ly.pi]), _f(self.params[ly.pi + 1]), _f(y), ly.cin, ly.cout, D, H, W,
                                 ly.ks, ly.stride)
                tape.append((ly, x))
            elif ly.kind == "conv_trans":
                y = np.empty((ly.cout, 2 * D, 2 * H, 2 * W), np.float32)
                l.orc_convt_fwd(_f(x), _f(self.params[ly.pi]), _f(self.params[ly.pi + 1]), _f(y), ly.cin, ly.cout, D, H, W)
                tape.append((ly, x))
            elif ly.kind in ("norm", "bnorm"):
                y = np.empty_like(x)
                S = D * H * W
                if ly.kind == "bnorm" and not self.training:
                    l.orc_bnorm_eval_fwd(_f(x), _f(self.params[ly.pi]), _f(self.params[ly.pi + 1]), _f(self.buffers[ly.bi]),
                                         _f(self.buffers[ly.bi + 1]), _f(y), Cc, C.c_int64(S), C.c_double(0.0))
                    tape.append((ly, None))
                else:
                    mean, rstd = np.empty(Cc, np.float32), np.empty(Cc, np.float32)
                    bn = ly.kind == "bnorm"
                    l.orc_norm_fwd(_f(x), _f(self.params[ly.pi]), _f(self.params[ly.pi + 1]), _f(y), _f(mean), _f(rstd), Cc,
                                   C.c_int64(S), C.c_double(0.0 if bn else 1e-5), _fn(self.buffers[ly.bi] if bn else None),
                                   _fn(self.buffers[ly.bi + 1] if bn else None), C.c_double(0.1))
                    if bn:
                        self.buffers[ly.bi + 2] += 1
                    tape.append((ly, (x, mean, rstd)))
            elif ly.kind == "act":
                y = np.empty_like(x)
                l.orc_act_fwd(_f(x), _f(y), C.c_int64(x.size), ly.act)
                tape.append((ly, x))
            elif ly.kind == "max_pool":
                y = np.empty((Cc, D // 2, H // 2, W // 2), np.float32)
                arg = np.empty(y.shape, np.int32)
                l.orc_maxpool_fwd(_f(x), _f(y), arg.ctypes.data_as(C.POINTER(C.c_int32)), Cc, D, H, W)
                tape.append((ly, (arg, x.shape)))
            elif ly.kind == "upsample":
                y = np.empty((Cc, 2 * D, 2 * H, 2 * W), np.float32)
                l.orc_upsample_fwd(_f(x), _f(y), Cc, D, H, W)
                tape.append((ly, x.shape))
            x = y
        return x

    def _seq_bwd(self, tape, g, grads):
        l = lib()
        for ly, saved in reversed(tape):
            if ly.kind == "conv":
                x = saved
                Cc, D, H, W = x.shape
                g = np.ascontiguousarray(g)
                l.orc_conv3d_bwd_weight(_f(x), _f(g), _f(grads[ly.pi]), _f(grads[ly.pi + 1]), ly.cin, ly.cout, D, H, W, ly.ks,
                                        ly.stride)
                dx = np.empty_like(x)
                l.orc_conv3d_bwd_data(_f(g), _f(self.params[ly.pi]), _f(dx), ly.cin, ly.cout, D, H, W, ly.ks, ly.stride)
            elif ly.kind == "conv_trans":
                x = saved
                Cc, D, H, W = x.shape
                g = np.ascontiguousarray(g)
                l.orc_convt_bwd_weight(_f(x), _f(g), _f(grads[ly.pi]), _f(grads[ly.pi + 1]), ly.cin, ly.cout, D, H, W)
                dx = np.empty_like(x)
                l.orc_convt_bwd_data(_f(g), _f(self.params[ly.pi]), _f(dx), ly.cin, ly.cout, D, H, W)
            elif ly.kind in ("norm", "bnorm"):
                x, mean, rstd = saved
                dx = np.empty_like(x)
                g = np.ascontiguousarray(g)
                l.orc_norm_bwd(_f(x), _f(g), _f(self.params[ly.pi]), _f(mean), _f(rstd), _f(dx), _f(grads[ly.pi]),
                               _f(grads[ly.pi + 1]), x.shape[0], C.c_int64(x[0].size))
            elif ly.kind == "act":
                dx = np.empty_like(saved)
                g = np.ascontiguousarray(g)
                l.orc_act_bwd(_f(saved), _f(g), _f(dx), C.c_int64(saved.size), ly.act)
            elif ly.kind == "max_pool":
                arg, shp = saved
                dx = np.empty(shp, np.float32)
                g = np.ascontiguousarray(g)
                l.orc_maxpool_bwd(_f(g), arg.ctypes.data_as(C.POINTER(C.c_int32)), _f(dx), shp[0], shp[1], shp[2], shp[3])
            elif ly.kind == "upsample":
                shp = saved
                dx = np.empty(shp, np.float32)
                g = np.ascontiguousarray(g)
                l.orc_upsample_fwd  # noqa (symmetry)
                l.orc_upsample_bwd(_f(g), _f(dx), shp[0], shp[1], shp[2], shp[3])
            g = dx
        return g

    def forward(self, x):
        """unet.cpp:168-193.  x: [Cin,D,H,W] float32.  Returns list of logits, keeps a tape for backward()."""
        x = np.ascontiguousarray(x, np.float32)
        n = len(self.encoding)
        self._tapes = {}
        skips = []
        for level in range(n):
            t = []
            x = self._seq_fwd(self.encoding[level], x, t)
            self._tapes[("e", level)] = t
            if level < n - 1:
                skips.append(x)
        results = [None] * len(self.output)
        self._skipc = [s.shape[0] for s in skips]
        for level in range(n - 2, -1, -1):
            x = np.ascontiguousarray(np.concatenate([skips[level], x], 0))
            skips[level] = None
            t = []
            x = self._seq_fwd(self.decoding[level], x, t)
            self._tapes[("d", level)] = t
            if self.output[level]:
                t = []
                results[level] = self._seq_fwd(self.output[level], x, t)
                self._tapes[("o", level)] = t
            if self.tail[level]:
                t = []
                x = self._seq_fwd(self.tail[level], x, t)
                self._tapes[("t", level)] = t
        return results

    def backward(self, dlogits, grads=None):
        """Reverse of forward(); accumulates into grads (list like self.params). Returns (grads, dinput)."""
        if grads is None:
            grads = [np.zeros_like(p) for p in self.params]
        n = len(self.encoding)
        g = None
        gskip = [None] * (n - 1)
        for level in range(0, n - 1):
            if self.tail[level]:
                g = self._seq_bwd(self._tapes[("t", level)], g, grads)
            if self.output[level] and dlogits[level] is not None:
                go = self._seq_bwd(self._tapes[("o", level)], np.ascontiguousarray(dlogits[level], np.float32), grads)
                g = go if g is None else g + go
            g = self._seq_bwd(self._tapes[("d", level)], g, grads)
            sc = self._skipc[level]
            gskip[level] = g[:sc]
            g = np.ascontiguousarray(g[sc:])
        for level in range(n - 1, -1, -1):
            if level < n - 1:
                g = g + gskip[level]
            g = self._seq_bwd(self._tapes[("e", level)], g, grads)
        return grads, g


def calc_losses(logits, target, Cn, collapse_before=0, weights=(1.0, 1.0, 1.0), want_grad=False):
    """train.cpp:501-552 -> ((ce,dice,mse), dlogits or None)."""
    if collapse_before < 0 or collapse_before >= Cn:
        raise RuntimeError("invalid collapse_before")
    logits = np.ascontiguousarray(logits, np.float32)
    target = np.ascontiguousarray(target, np.int64)
    S = target.size
    out = (C.c_double * 3)()
    dl = np.empty_like(logits) if want_grad else None
    lib().orc_calc_losses(_f(logits), target.ctypes.data_as(C.POINTER(C.c_int64)), Cn, C.c_int64(S), collapse_before,
                          C.c_double(weights[0]), C.c_double(weights[1]), C.c_double(weights[2]), out, _fn(dl))
    return (out[0], out[1], out[2]), dl


def target_half(t):
    """train.cpp:645-662."""
    t = np.ascontiguousarray(t, np.int64)
    D, H, W = t.shape
    o = np.empty((D >> 1, H >> 1, W >> 1), np.int64)
    if o.size == 0:
        raise RuntimeError("deep supervision target size became zero")
    p = C.POINTER(C.c_int64)
    lib().orc_target_half(t.ctypes.data_as(p), o.ctypes.data_as(p), D, H, W)
    return o


def deep_supervision(outputs, target, out_count, cost=(True, True, True), collapse_before=0):
    """train.cpp:634-706 -> (total_loss, (ce,dice,mse) at level 0, [dlogits per level])."""
    ws = sum(1.0 / (1 << k) for k in range(len(outputs)))
    total, stats, grads = 0.0, None, []
    act = np.asarray(target, np.int64)
    w3 = tuple(1.0 if c else 0.0 for c in cost)
    if not any(cost):
        w3 = (1.0, 0.0, 0.0)  # train.cpp:696-697
    for k, o in enumerate(outputs):
        if k > 0:
            act = target_half(act)
        if o is None:
            raise RuntimeError("undefined deep supervision output at level %d" % k)
        nw = np.float32(np.float32(1.0 / (1 << k)) * np.float32(1.0 / np.float32(ws)))
        (ce, dice, mse), dl = calc_losses(o, act, out_count, collapse_before, tuple(w * float(nw) for w in w3), True)
        if k == 0:
            stats = (ce, dice, mse)
        total += float(nw) * (w3[0] * ce + w3[1] * dice + w3[2] * mse)
        grads.append(dl)
    return total, stats, grads


def decay_mask(model):
    """unet.cpp:250-259: weight decay 3e-5 on params with dim > 1 (bias-named and 1-D params get 0)."""
    return [p.ndim > 1 for p in model.params]


def step_epilogue(model, grads, mom, batch_size, lr, first_step, max_norm=12.0, momentum=0.99, wd=3e-5):
    """train.cpp:759-766 with the optimizer of unet.cpp:246-277.  Returns the pre-clip grad norm."""
    l = lib()
    tot = 0.0
    for g in grads:
        g /= np.float32(batch_size)
        tot += l.orc_grad_norm(_f(g), C.c_int64(g.size)) ** 2
    norm = float(np.sqrt(tot))
    coef = min(1.0, max_norm / (norm + 1e-6))
    for p, g, m, d in zip(model.params, grads, mom, decay_mask(model)):
        l.orc_sgd_step(_f(p), _f(g), _f(m), C.c_int64(p.size), C.c_double(lr), C.c_double(momentum),
                       C.c_double(wd if d else 0.0), C.c_double(coef), int(first_step))
    return norm
