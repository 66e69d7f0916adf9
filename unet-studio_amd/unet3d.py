"""Python host-side mirror of the reference's UNet3d operator surface (unet.hpp:13-70, unet.cpp).

Same names, argument meaning and error behaviour as `UNet3dImpl`; torch is used only for device memory,
streams and (in train.py) torch.distributed.  All arithmetic happens in libunet_hip.so through the
C ABI of include/unet_hip.h.
"""
import math
import threading

import torch

from . import engine as E

_DT = {"fp32": E.DTYPE_F32, "f32": E.DTYPE_F32, "float32": E.DTYPE_F32, "bf16": E.DTYPE_BF16, "bfloat16": E.DTYPE_BF16}


_NO_FUSED_LOSS = False   # (forward and loss as two engine calls: test_fused_forward_loss_equals_the_two_calls compares them through the ABI)


def _stream_ptr(device):
    return torch.cuda.current_stream(device).cuda_stream


class SGD:
    """What callers use of torch::optim::SGD (train.cpp:567-571,765-766): param_groups()[i].lr, step(), zero_grad().
    Configuration is create_optimizer's (unet.cpp:246-277): momentum 0.99, nesterov, weight decay 3e-5 on
    parameters with dim > 1 and no "bias" in the name."""

    def __init__(self, model, lr):
        self.model = model
        self.param_groups = [{"lr": lr, "weight_decay": 3e-5}, {"lr": lr, "weight_decay": 0.0}]
        self.momentum, self.nesterov, self.weight_decay = 0.99, True, 3e-5
        self.momentum_buffer = torch.zeros_like(model.flat_params)
        self.last_grad_norm = torch.zeros(1, device=model.flat_params.device, dtype=torch.float32)
        self._scratch = torch.empty(65536, dtype=torch.uint8, device=model.flat_params.device)

    def step(self, grad_scale=1.0, clip_norm=12.0):
        """grad /= batch_size (grad_scale), clip_grad_norm_(12.0), SGD step, zero_grad -- train.cpp:759-766 in one pass."""
        m = self.model
        lr = self.param_groups[0]["lr"]
        plan = m._any_plan()
        E.check(E.lib.unet_sgd_step(plan.handle, m.flat_params.data_ptr(), m.flat_grads.data_ptr(),
                                    self.momentum_buffer.data_ptr(), lr, self.momentum, int(self.nesterov),
                                    self.weight_decay, clip_norm, grad_scale, self.last_grad_norm.data_ptr(),
                                    self._scratch.data_ptr(), _stream_ptr(m.device())))
        m._params_version += 1

    def zero_grad(self):
        self.model.flat_grads.zero_()

    def state_dict(self):
        return {"momentum_buffer": self.momentum_buffer.clone(), "lr": self.param_groups[0]["lr"]}

    def load_state_dict(self, sd):
        if sd["momentum_buffer"].numel() != self.momentum_buffer.numel():
            raise E.UNetError("optimizer state does not belong to this architecture: %d momentum elements, %d parameters"
                              % (sd["momentum_buffer"].numel(), self.momentum_buffer.numel()))
        self.momentum_buffer.copy_(sd["momentum_buffer"].to(self.momentum_buffer.device))
        for g in self.param_groups:
            g["lr"] = sd["lr"]


class _Forward(torch.autograd.Function):
    """Lets torch losses drive backward() (total_loss.backward(), train.cpp:706)."""

    @staticmethod
    def forward(ctx, x, trigger, model, plan, ws):
        outs = model._run_forward(plan, ws, x, mode=1)
        ctx.model, ctx.plan, ctx.ws = model, plan, ws
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        gouts = [g.contiguous() if g is not None else None for g in gouts]
        ctx.model._run_backward(ctx.plan, ctx.ws, gouts)
        return None, None, None, None, None


class UNet3d:
    """UNet3d(in_count, out_count, architecture) -- unet.hpp:45-46.  Raises engine.UNetError (the C++ host:
    std::runtime_error) for DSL errors, like unet.cpp:53,66,88,117."""

    def __init__(self, in_count, out_count, architecture, device="cuda:0", dtype="bf16", impl=E.IMPL_AUTO, seed=None):
        self.in_count, self.out_count, self.architecture = int(in_count), int(out_count), architecture
        # unet.cpp:110-112 and unet.hpp:37-38 defaults
        self.fov_strategy, self.preproc, self.postproc = "align_top", "", "softmax+create_mask+argmax"
        self.orientation, self.error_msg = "", ""
        self.voxel_size, self.dim = (1.0, 1.0, 1.0), (192, 224, 192)
        self.testing_errors, self.training_errors, self.single_component_label = [], [], []
        self.error_mutex = threading.Lock()
        self.optimizer = None
        self._device = torch.device(device)
        self._dtype, self._impl = _DT[dtype], impl
        self._training = True
        self._plans, self._workspaces = {}, {}
        self._params_version = 0
        # a structural plan (size-independent facts: parameter shapes/order); also validates the DSL
        probe = E.Plan(architecture, self.in_count, self.out_count, self._probe_size(architecture), self._dtype,
                       self._device.index or 0, impl)
        self._probe = probe
        self.param_shapes = probe.param_shapes
        sizes = [int(math.prod(s)) for s in self.param_shapes]
        self._offsets = [0]
        for n in sizes:
            self._offsets.append(self._offsets[-1] + n)
        self.flat_params = torch.zeros(self._offsets[-1], dtype=torch.float32, device=self._device)
        self.flat_grads = torch.zeros_like(self.flat_params)
        self._params = [self.flat_params[self._offsets[i]:self._offsets[i + 1]].view(self.param_shapes[i])
                        for i in range(len(sizes))]
        self._grads = [self.flat_grads[self._offsets[i]:self._offsets[i + 1]].view(self.param_shapes[i])
                       for i in range(len(sizes))]
        self._buffers = [torch.zeros(n, dtype=torch.float32, device=self._device) for n in probe.buffer_numel]
        for i in range(1, len(self._buffers), 2):
            self._buffers[i].fill_(1.0)  # running_var
        self.num_batches_tracked = 0
        # host arrays of device pointers (unet_forward / unet_backward arguments): the flat buffers never move, build them once
        self._pp = E.ptr_array([p.data_ptr() for p in self._params])
        self._gp = E.ptr_array([g.data_ptr() for g in self._grads])
        self._bp = E.ptr_array([b.data_ptr() for b in self._buffers]) if self._buffers else None
        self._trigger = torch.zeros(1, device=self._device, requires_grad=True)
        self.reset_parameters(seed)

    @staticmethod
    def _probe_size(arch):
        n = sum(1 for l in arch.replace("\r", "").split("\n") if l.strip())
        s = 1 << max(n, 3)  # large enough that no level collapses to zero for any legal DSL of that depth
        return (s, s, s)

    # ---- torch::nn::Module surface used by the callers ----
    def parameters(self):
        return list(self._params)

    def grads(self):
        return list(self._grads)

    def buffers(self):
        return list(self._buffers)

    def device(self):
        return self._device

    def to(self, device):
        if torch.device(device) != self._device:
            raise E.UNetError("a UNet3d lives on the device it was constructed on: build one on %s and copy_from()" % device)
        return self

    def train(self, on=True):
        """unet.hpp:53-62."""
        self._training = bool(on)
        return self

    def eval(self):
        return self.train(False)

    def set_requires_grad(self, req):
        self._training = self._training and bool(req)

    def reset_parameters(self, seed=None):
        """libtorch default init: conv / conv_trans weight and bias U(-1/sqrt(fan_in), 1/sqrt(fan_in))
        (kaiming_uniform_(a=sqrt(5))), norm weight 1, bias 0."""
        g = torch.Generator(device="cpu")
        if seed is not None:
            g.manual_seed(seed)
        with torch.no_grad():
            for p, fan, isn in zip(self._params, self._probe.param_fan_in, self._probe.param_is_norm_weight):
                if fan > 0:
                    b = 1.0 / math.sqrt(fan)
                    p.copy_(((torch.rand(p.shape, generator=g) * 2 - 1) * b).to(self._device))
                else:
                    p.fill_(1.0 if isn else 0.0)
        self._params_version += 1

    def load_parameters(self, arrays, buffers=None):
        """tensor<i> in parameters() order (main.cpp:193-204)."""
        assert len(arrays) == len(self._params), "parameter count mismatch"
        with torch.no_grad():
            for p, a in zip(self._params, arrays):
                a = torch.as_tensor(a, dtype=torch.float32)
                if a.numel() != p.numel():
                    raise E.UNetError("tensor size mismatch")  # main.cpp:198-200
                p.copy_(a.reshape(p.shape).to(self._device))
            if buffers is not None:
                for b, a in zip(self._buffers, buffers):
                    b.copy_(torch.as_tensor(a, dtype=torch.float32).reshape(b.shape).to(self._device))
        self._params_version += 1

    def copy_from(self, r):
        """unet.cpp:195-222."""
        with torch.no_grad():
            for a, b in zip(self._params, r._params):
                if a.shape == b.shape:
                    a.copy_(b)
            for a, b in zip(self._buffers, r._buffers):
                if a.shape == b.shape:
                    a.copy_(b)
        self.voxel_size, self.dim = r.voxel_size, r.dim
        self.fov_strategy, self.postproc, self.preproc = r.fov_strategy, r.postproc, r.preproc
        self._params_version += 1

    def add_gradient_from(self, r):
        """unet.cpp:224-244 (the in-process reduce-to-root; across ranks train.py all-reduces instead)."""
        self.flat_grads.add_(r.flat_grads.to(self._device, torch.float32))

    def create_optimizer(self, learning_rate):
        """unet.cpp:246-277."""
        self.optimizer = SGD(self, learning_rate)
        return self.optimizer

    def save_optimizer(self, file_name):
        """torch::save(*(model->optimizer), model_path + ".opt") -- train.cpp:787: the momentum of the fused update (and the learning
        rate), so that a run resumed from <model>.nz + <model>.nz.opt continues exactly.  Returns False + error_msg on failure.
        FORMAT: a torch.save pickle {"momentum_buffer": flat fp32 tensor, "lr": float} -- the Python host's own.  It is NOT the libtorch
        optimizer archive that the C++ host (unet_host.cpp:save_optimizer) and the reference (train.cpp:787) write with
        torch::save(*optimizer): a .opt file of one host cannot be loaded by the other (load_optimizer reports "cannot load optimizer")."""
        try:
            if self.optimizer is None:
                raise E.UNetError("save_optimizer: no optimizer (create_optimizer first)")
            sd = self.optimizer.state_dict()
            torch.save({"momentum_buffer": sd["momentum_buffer"].cpu(), "lr": sd["lr"]}, file_name)
            return True
        except (OSError, RuntimeError) as e:
            self.error_msg = "cannot save optimizer: %s" % e
            return False

    def load_optimizer(self, file_name):
        """torch::load(*(model->optimizer), model_path + ".opt") -- train.cpp:945-957 ("cannot load optimizer: ..." on failure)."""
        try:
            if self.optimizer is None:
                raise E.UNetError("load_optimizer: no optimizer (create_optimizer first)")
            self.optimizer.load_state_dict(torch.load(file_name, map_location="cpu"))
            return True
        except (OSError, RuntimeError, KeyError) as e:
            self.error_msg = "cannot load optimizer: %s" % e
            return False

    def prepare_for_inference(self, device=None):
        """unet.cpp:7-22: eval() and running_mean 0 / running_var 1 / num_batches_tracked 0 on every BatchNorm3d."""
        self.eval()
        for i in range(0, len(self._buffers), 2):
            self._buffers[i].zero_()
            self._buffers[i + 1].fill_(1.0)
        self.num_batches_tracked = 0

    def get_training_errors(self):
        with self.error_mutex:
            return list(self.training_errors)

    def get_testing_errors(self):
        with self.error_mutex:
            return list(self.testing_errors)

    def get_info(self):
        """unet.cpp:279-291."""
        s = "in: %d out: %d\n" % (self.in_count, self.out_count)
        s += "dim: %s reso: %s\n" % (" ".join(str(d) for d in self.dim), " ".join(str(v) for v in self.voxel_size))
        s += "structure: %s\n" % self.architecture
        if self.preproc:
            s += "preproc: %s\n" % self.preproc
        if self.postproc:
            s += "postproc: %s\n" % self.postproc
        return s

    def print_layers(self):
        print(self._probe.describe())

    # ---- plans and workspaces ----
    def plan_for(self, size):
        key = tuple(int(s) for s in size)
        p = self._plans.get(key)
        if p is None:
            p = E.Plan(self.architecture, self.in_count, self.out_count, key, self._dtype, self._device.index or 0, self._impl)
            self._plans[key] = p
        return p

    def _any_plan(self):
        return next(iter(self._plans.values())) if self._plans else self._probe

    def _workspace(self, plan):
        """one workspace per (plan, host thread): forward is re-entrant across threads (qc.cpp:273-297)"""
        key = (plan.size, threading.get_ident())
        ws = self._workspaces.get(key)
        if ws is None:
            ws = torch.empty(plan.workspace_bytes, dtype=torch.uint8, device=self._device)
            self._workspaces[key] = ws
        return ws

    def _run_forward(self, plan, ws, x, mode):
        outs = [torch.empty(s, dtype=torch.float32, device=self._device) if s[1] > 0 else None for s in plan.output_shapes]
        pp, bp = self._pp, self._bp
        op = E.ptr_array([o.data_ptr() if o is not None else None for o in outs])
        E.check(E.lib.unet_forward(plan.handle, pp, bp, x.data_ptr(), op, ws.data_ptr(), mode, _stream_ptr(self._device)))
        if (mode & 1) and self._buffers:
            self.num_batches_tracked += 1
        return outs

    def pack_filters(self, size, with_dgrad=True):
        """the filter packs of the workspace for volumes of `size`, made now on the current stream (unet_pack_filters); True when the
        next forward at this size may be given packs_current"""
        import ctypes
        plan = self.plan_for(tuple(size))
        ws = self._workspace(plan)
        made = ctypes.c_int(0)
        E.check(E.lib.unet_pack_filters(plan.handle, self._pp, ws.data_ptr(), 1 if with_dgrad else 0, ctypes.byref(made),
                                        _stream_ptr(self._device)))
        return bool(made.value)

    def _run_backward(self, plan, ws, grad_outs, grad_x=None, gp=None):
        """gp: pointer array of a gradient buffer other than flat_grads (grad_pointers), for micro-steps that run side by side"""
        pp, gp = self._pp, (gp if gp is not None else self._gp)
        go = E.ptr_array([g.data_ptr() if g is not None else None for g in grad_outs])
        E.check(E.lib.unet_backward(plan.handle, pp, go, gp, grad_x.data_ptr() if grad_x is not None else None,
                                    ws.data_ptr(), _stream_ptr(self._device)))

    def _run_backward_part(self, plan, ws, grad_outs, op_hi, op_lo):
        pp, gp = self._pp, self._gp
        go = E.ptr_array([g.data_ptr() if g is not None else None for g in grad_outs])
        E.check(E.lib.unet_backward_part(plan.handle, pp, go, gp, None, ws.data_ptr(), op_hi, op_lo, _stream_ptr(self._device)))

    def forward_backward_bucketed(self, x, target, on_bucket, cost_ce=True, cost_dice=True, cost_mse=True, collapse_before=0,
                                  max_buckets=3, packs_current=False, losses_out=None):
        """forward_backward with the backward issued in buckets: on_bucket(elem_lo, elem_hi) is called after each part, when the
        gradients flat_grads[elem_lo:elem_hi] are final on the current stream (the data-parallel trainer starts their all-reduce
        there, so that it runs under the rest of the backward)."""
        x = self._check_input(x)
        plan = self.plan_for(x.shape[2:])
        ws = self._workspace(plan)
        outs, losses, gouts = self._run_forward_loss(plan, ws, x, target, cost_ce, cost_dice, cost_mse, collapse_before, packs_current,
                                                     losses_out=losses_out)
        op_hi, elem_hi = 1 << 30, int(self.flat_grads.numel())
        for op_lo, elem_lo in plan.backward_buckets(max_buckets):
            self._run_backward_part(plan, ws, gouts, op_hi, op_lo)
            on_bucket(int(elem_lo), elem_hi)
            op_hi, elem_hi = op_lo, int(elem_lo)
        return losses

    def grad_pointers(self, flat):
        """host array of device pointers (unet_backward's grad_params) into another flat fp32 buffer laid out like flat_grads"""
        assert flat.numel() == self.flat_grads.numel() and flat.dtype == torch.float32 and flat.device == self._device
        return E.ptr_array([flat.data_ptr() + 4 * self._offsets[i] for i in range(len(self._params))])

    def make_lane(self, size, cu_range=None):
        """What a micro-step needs for ITSELF to run beside another one of the same optimizer step: a stream, a plan (a plan owns the
        side stream and the events of its backward: two backwards on ONE plan at a time are not supported, include/unet_hip.h), a
        workspace and a loss scratch.  The reference runs the batch_size micro-steps of a step on as many threads as it has GPUs
        (train.cpp:581-606); with one GPU per process the same independence lets two samples share the device, one sample's latency-
        bound small levels under the other's bandwidth-bound large ones."""
        key = tuple(int(v) for v in size)
        plan = E.Plan(self.architecture, self.in_count, self.out_count, key, self._dtype, self._device.index or 0, self._impl)
        stream = torch.cuda.Stream(self._device)
        if cu_range is not None:    # experiment: the lane (its stream and its plan's side stream) on CUs [first, first + count) of every XCD
            import ctypes
            h = ctypes.c_void_p()
            E.check(E.lib.unet_stream_create_cu_range(self._device.index or 0, int(cu_range[0]), int(cu_range[1]), ctypes.byref(h)))
            stream = torch.cuda.ExternalStream(h.value, device=self._device)
            E.check(E.lib.unet_plan_side_cu_range(plan.handle, int(cu_range[0]), int(cu_range[1])))
        return {"plan": plan, "stream": stream,
                "ws": torch.empty(plan.workspace_bytes, dtype=torch.uint8, device=self._device),
                "loss": torch.empty(plan.loss_scratch_bytes, dtype=torch.uint8, device=self._device)}

    def forward_backward_lane(self, lane, x, target, gp, cost_ce=True, cost_dice=True, cost_mse=True, collapse_before=0, packs_current=False):
        """forward_backward on a lane (make_lane), on the CURRENT stream, gradients accumulated through the pointer array gp"""
        x = self._check_input(x)
        plan, ws = lane["plan"], lane["ws"]
        outs, losses, gouts = self._run_forward_loss(plan, ws, x, target, cost_ce, cost_dice, cost_mse, collapse_before, packs_current,
                                                     loss_scratch=lane["loss"])
        self._run_backward(plan, ws, gouts, gp=gp)
        return losses

    def _check_input(self, x):
        if x.dim() != 5 or x.size(0) != 1 or x.size(1) != self.in_count:
            raise E.UNetError("forward expects a {1,%d,D,H,W} tensor, got %s" % (self.in_count, tuple(x.shape)))
        if x.device != self._device:
            raise E.UNetError("input is on %s, model is on %s" % (x.device, self._device))
        return x.to(torch.float32).contiguous()

    def forward(self, x, packs_current=False):
        """std::vector<torch::Tensor> forward(torch::Tensor) -- unet.hpp:51, unet.cpp:168-193.
        x: {1,in_count,D,H,W} fp32 on device(); returns one fp32 tensor per decoder level, [0] full resolution.
        In train() mode (and with grad enabled) the outputs carry an autograd edge whose backward
        accumulates into grads()."""
        x = self._check_input(x)
        plan = self.plan_for(x.shape[2:])
        ws = self._workspace(plan)
        if self._training and torch.is_grad_enabled():
            return list(_Forward.apply(x, self._trigger, self, plan, ws))
        # packs_current (eval loops only): the caller asserts that an eval forward of THIS size has run on this thread since the parameters
        # last changed, so the filter packs in the workspace are current (UNET_MODE_PACKS_CURRENT)
        return self._run_forward(plan, ws, x, mode=(1 if self._training else 0) | (E.MODE_PACKS_CURRENT if packs_current else 0))

    __call__ = forward

    # ---- fused train micro-step (train.cpp:615-706 for one sample) ----
    def forward_backward(self, x, target, cost_ce=True, cost_dice=True, cost_mse=True, collapse_before=0, packs_current=False, losses_out=None):
        """forward + calc_losses over all deep-supervision levels + backward, all in the engine.
        target: int64 {1,D,H,W}.  Returns a device tensor {total, ce0, dice0, mse0}; gradients are ACCUMULATED
        into grads().  packs_current: the caller asserts that a training forward has run on this thread's workspace since the
        parameters last changed (micro-steps 2.. of one optimizer step): the engine skips the filter repack.  losses_out: a device
        tensor of 4 floats to receive {total, ce0, dice0, mse0} instead of a fresh one."""
        x = self._check_input(x)
        plan = self.plan_for(x.shape[2:])
        ws = self._workspace(plan)
        outs, losses, gouts = self._run_forward_loss(plan, ws, x, target, cost_ce, cost_dice, cost_mse, collapse_before, packs_current,
                                                     losses_out=losses_out)
        self._run_backward(plan, ws, gouts)
        return losses

    def _loss_scratch(self, plan):
        key = ("loss", plan.size, threading.get_ident())
        sc = self._workspaces.get(key)
        if sc is None:
            sc = torch.empty(plan.loss_scratch_bytes, dtype=torch.uint8, device=self._device)
            self._workspaces[key] = sc
        return sc

    def _run_forward_loss(self, plan, ws, x, target, cost_ce, cost_dice, cost_mse, collapse_before, packs_current=False, loss_scratch=None,
                          losses_out=None):
        """train-mode forward + calc_losses over all levels in ONE engine call (unet_forward_loss): same numbers as forward() then
        loss(), with the coarse levels' loss kernels issued beside the rest of the decoder."""
        if target.dtype != torch.int64 or target.device != self._device:
            raise E.UNetError("target must be an int64 tensor on the model's device")
        target = target.contiguous()
        if any(s[1] == 0 for s in plan.output_shapes) or _NO_FUSED_LOSS:   # a level without a head: the two-call path reports it (train.cpp:664-671)
            outs = self._run_forward(plan, ws, x, mode=1 | (E.MODE_PACKS_CURRENT if packs_current else 0))
            losses, gouts = self.loss(outs, target, cost_ce, cost_dice, cost_mse, collapse_before, plan=plan)
            if losses_out is not None:
                losses_out.copy_(losses)
                losses = losses_out
            return outs, losses, gouts
        outs = [torch.empty(s, dtype=torch.float32, device=self._device) for s in plan.output_shapes]
        gouts = [torch.empty_like(o) for o in outs]
        losses = losses_out if losses_out is not None else torch.empty(4, dtype=torch.float32, device=self._device)
        mask = (1 if cost_ce else 0) | (2 if cost_dice else 0) | (4 if cost_mse else 0)
        mode = 1 | (E.MODE_PACKS_CURRENT if packs_current else 0)
        E.check(E.lib.unet_forward_loss_mode(plan.handle, self._pp, self._bp, x.data_ptr(), E.ptr_array([o.data_ptr() for o in outs]),
                                             target.data_ptr(), mask, collapse_before, E.ptr_array([g.data_ptr() for g in gouts]),
                                             losses.data_ptr(), (loss_scratch if loss_scratch is not None else self._loss_scratch(plan)).data_ptr(),
                                             ws.data_ptr(), mode,
                                             _stream_ptr(self._device)))
        if self._buffers:
            self.num_batches_tracked += 1
        return outs, losses, gouts

    def loss(self, outs, target, cost_ce=True, cost_dice=True, cost_mse=True, collapse_before=0, want_grad=True, plan=None):
        """calc_losses + deep-supervision weighting (train.cpp:501-552,634-706) -> (losses[4], dL/d(outs))."""
        if plan is None:
            plan = self.plan_for(outs[0].shape[2:])
        if target.dtype != torch.int64 or target.device != self._device:
            raise E.UNetError("target must be an int64 tensor on the model's device")
        target = target.contiguous()
        sc = self._loss_scratch(plan)
        gouts = [torch.empty_like(o) if (o is not None and want_grad) else None for o in outs]
        losses = torch.empty(4, dtype=torch.float32, device=self._device)
        mask = (1 if cost_ce else 0) | (2 if cost_dice else 0) | (4 if cost_mse else 0)
        E.check(E.lib.unet_loss(plan.handle, E.ptr_array([o.data_ptr() if o is not None else None for o in outs]),
                                target.data_ptr(), mask, collapse_before,
                                E.ptr_array([g.data_ptr() if g is not None else None for g in gouts]),
                                losses.data_ptr(), sc.data_ptr(), _stream_ptr(self._device)))
        return losses, gouts
