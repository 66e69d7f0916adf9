"""Host-side mirror of the reference's training step (train.hpp:8-30, train.cpp:554-805), data parallel.

One process per GPU.  The `batch_size` single-volume micro-steps of one optimizer step ("epoch",
train.cpp:562,778) are split statically over the ranks (b % world_size == rank), which is equivalent to
the reference's dynamic stealing (train.cpp:604-606) because the step sums all sample gradients before
use (train.cpp:756-761).  The reference's reduce-to-root + per-step broadcast (unet.cpp:224-244,
train.cpp:573-579) becomes ONE sum all-reduce of the flat fp32 gradient buffer over RCCL/xGMI, after
which every rank applies the identical update -- no broadcast needed.
"""
import math
import os

import torch
import torch.distributed as dist


class TrainingParam:
    """training_param, train.hpp:8-30 (the fields the step loop reads)."""

    def __init__(self, batch_size=32, epoch=10000, learning_rate=0.001, seed=0, cost_ce=True, cost_dice=True, cost_mse=True):
        self.batch_size, self.epoch, self.learning_rate, self.seed = batch_size, epoch, learning_rate, seed
        self.cost_ce, self.cost_dice, self.cost_mse = cost_ce, cost_dice, cost_mse


class SyntheticVolumes:
    """Stands in for the reader/augmentation threads (train.cpp:267-485, out of scope): seeded synthetic
    samples born on the device.  image U[0,1) fp32 {1,in,D,H,W}; label U{0..out-1} int64 {1,D,H,W};
    seed = sample index, like in_file_seed (train.cpp:439)."""

    def __init__(self, in_count, out_count, size, device, base_seed=0, cache=0):
        self.in_count, self.out_count, self.size, self.device, self.base_seed = in_count, out_count, tuple(size), device, base_seed
        self._cache, self._cache_n = {}, cache

    def __call__(self, index):
        if index in self._cache:
            return self._cache[index]
        g = torch.Generator(device=self.device)
        g.manual_seed(self.base_seed + index)
        x = torch.rand((1, self.in_count) + self.size, generator=g, device=self.device, dtype=torch.float32)
        t = torch.randint(0, self.out_count, (1,) + self.size, generator=g, device=self.device, dtype=torch.int64)
        if len(self._cache) < self._cache_n:
            self._cache[index] = (x, t)
        return x, t


class Trainer:
    """train_unet's thread C (train.cpp:554-805) for one rank."""

    def __init__(self, model, param, source, rank=0, world_size=1, group=None, comm=None):
        """comm: an engine.Comm (RCCL under the C ABI, include/unet_hip.h unet_comm_*): the gradient / statistics collectives then
        go through unet_allreduce_grads instead of torch.distributed -- the same calls the C++ host makes (unet_host.cpp)."""
        self.model, self.param, self.source = model, param, source
        self.rank, self.world_size, self.group, self.comm = rank, world_size, group, comm
        if comm is not None:
            self.rank, self.world_size = comm.rank, comm.world
        # world_size > 1: start the all-reduce of every finished gradient bucket under the rest of the backward
        self.overlap = True        # attribute (tests set it False: one all-reduce after the backward)
        # micro-steps 2.. of a step skip the filter repack (UNET_MODE_PACKS_CURRENT); models without the keyword (test stand-ins) do not
        self.packs_reuse = hasattr(model, "_run_forward_loss")
        # micro-steps of one optimizer step in flight on this GPU at a time (UNET_MICRO_IN_FLIGHT; default 1 = one after the other:
        # measured on MI355X at 128^3 bf16, batch 8: 2.99 ms per sample sequentially, 3.76 ms with two in flight -- kernels of two
        # samples that share the chip thrash each other's L2 patches and LDS occupancy)
        self.in_flight = max(1, int(os.environ.get("UNET_MICRO_IN_FLIGHT", "1")))
        # the engine writes the first micro-step's {total, ce, dice, mse} straight into the step's statistics (no zeroing launch, no add)
        # pack_after_update (attribute, default off: measured neutral): the filter packs are made right behind the update, on the caller's
        # stream, instead of by the next step's forward on the side stream beside its first kernels
        self.pack_after_update = False
        self._packed_size, self._packed_version = None, None
        self.stats_direct = hasattr(model, "_run_forward_loss")
        self._lanes, self._gbufs, self._gptrs = None, [], []
        self.cur_epoch = 0
        model.train()
        if model.optimizer is None:
            model.create_optimizer(param.learning_rate)  # train.cpp:942
        self._stats = torch.zeros(4, dtype=torch.float32, device=model.device())

    def lr_at(self, epoch):
        """train.cpp:566."""
        return self.param.learning_rate * math.pow(1.0 - float(epoch) / self.param.epoch, 0.9)

    def step(self):
        """one optimizer step: train.cpp:562-789 without validation / checkpoint I/O"""
        p, m = self.param, self.model
        cur_data_index = self.cur_epoch * p.batch_size
        for g in m.optimizer.param_groups:
            g["lr"] = self.lr_at(self.cur_epoch)
        count = 0
        mine = list(range(self.rank, p.batch_size, self.world_size))
        works = []
        # The collective sequence must be a pure function of (plan, world_size, batch_size): a rank without a sample of this step
        # (batch_size < world_size; the reference simply starts min(gpus, batch_size) threads, train.cpp:581-582) never enters the
        # bucketed backward, so the per-bucket all-reduces are only used when EVERY rank has at least one micro-step.
        multi = self.world_size > 1 or self.comm is not None
        overlap = (multi and self.overlap and hasattr(m, "forward_backward_bucketed") and p.batch_size >= self.world_size)
        stream = torch.cuda.current_stream(m.device()).cuda_stream if self.comm is not None else None

        def reduce_bucket(lo, hi):   # the bucket's gradients are final: sum them over the replicas under the rest of the backward
            if hi > lo and self.comm is not None:
                self.comm.allreduce(m.flat_grads, lo, hi, stream)     # on the communicator's stream, after the bucket's kernels
            elif hi > lo:
                works.append(dist.all_reduce(m.flat_grads[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

        # Two micro-steps of this rank side by side (item: the reference's batch_size micro-steps are independent until the gradient
        # sum, train.cpp:604-606,756-761).  Each writes a gradient buffer of its own; unet_sum_buffers adds them in micro-step order, which
        # is exactly what one accumulating buffer would hold, so the update is bit-identical to the sequential order.  Not with bnorm
        # (the running statistics are updated in forward order) and not for models without lanes (test stand-ins).
        lanes_on = (self.in_flight > 1 and len(mine) > 1 and hasattr(m, "make_lane") and not m.buffers())
        if lanes_on or not mine or not self.stats_direct:
            self._stats.zero_()
        if lanes_on:
            losses_all = self._step_in_lanes(mine, cur_data_index)
            for l in losses_all:
                self._stats += l
            count = len(mine)
            mine = []
            overlap = False
        # workspaces (and the filter packs they hold) are keyed by the sample's spatial size: a micro-step may only skip the repack when a
        # micro-step of THIS optimizer step (or unet_pack_filters behind the last update) already packed for its size
        packed_sizes = set()
        if self._packed_size is not None and self._packed_version == getattr(m, "_params_version", None):
            packed_sizes.add(self._packed_size)
        for k, b in enumerate(mine):
            x, t = self.source(cur_data_index + b)
            # the parameters only change at the end of the step (train.cpp:765): this rank's micro-steps 2.. reuse the filter packs of its first
            last_size = tuple(x.shape[2:])
            kw = {"packs_current": True} if (self.packs_reuse and last_size in packed_sizes) else {}
            packed_sizes.add(last_size)
            first_direct = k == 0 and self.stats_direct
            if first_direct:
                kw["losses_out"] = self._stats
            if overlap and k == len(mine) - 1:
                # gradients accumulate over this rank's micro-steps: only the last backward can hand finished buckets to RCCL
                losses = m.forward_backward_bucketed(x, t, reduce_bucket, p.cost_ce, p.cost_dice, p.cost_mse, **kw)
            else:
                losses = m.forward_backward(x, t, p.cost_ce, p.cost_dice, p.cost_mse, **kw)
            if not first_direct:
                self._stats += losses
            count += 1
        if self.comm is not None:
            if not overlap:
                self.comm.allreduce(m.flat_grads, 0, m.flat_grads.numel(), stream)
            self.comm.allreduce(self._stats, 0, 4, stream)          # loss-stat gather, train.cpp:732-741
            self.comm.join(stream)                                  # the update below reads the summed gradients
        elif self.world_size > 1:
            # gradient sum over replicas (unet.cpp:224-244 -> RCCL all-reduce of the flat buffer, in buckets when overlapped)
            if not overlap:
                dist.all_reduce(m.flat_grads, op=dist.ReduceOp.SUM, group=self.group)
            dist.all_reduce(self._stats, op=dist.ReduceOp.SUM, group=self.group)  # loss-stat gather, train.cpp:732-741
            for w in works:
                w.wait()
        m.optimizer.step(grad_scale=1.0 / p.batch_size, clip_norm=12.0)  # train.cpp:759-766
        self._packed_size = None
        if self.pack_after_update and mine and not lanes_on and m.pack_filters(last_size):
            self._packed_size, self._packed_version = last_size, m._params_version
        if self.comm is not None and self.world_size > 1:
            for b in m.buffers():
                self.comm.broadcast(b, 0, stream)
            self.comm.join(stream)
        elif self.world_size > 1:
            # bnorm running statistics: the reference overwrites every replica's buffers with the root's each step (copy_from,
            # unet.cpp:207-215, train.cpp:573-579), so only rank 0's statistics exist -- keep it so (validate / save on any rank)
            for b in (m.buffers() if hasattr(m, "buffers") else []):
                dist.broadcast(b, 0, group=self.group)
        self.cur_epoch += 1
        return self._stats

    def _step_in_lanes(self, mine, cur_data_index):
        """this rank's micro-steps of one optimizer step, `in_flight` at a time: micro-step k on lane k % in_flight (stream order inside a
        lane), gradients of micro-step k into buffer k (buffer 0 = flat_grads), then ONE ordered sum into flat_grads"""
        import torch.cuda as tc
        from . import engine as E
        p, m = self.param, self.model
        x0, _ = self.source(cur_data_index + mine[0])
        if self._lanes is None:
            # UNET_LANE_CUS=<CUs per XCD and lane> (experiment): lane i on CUs [i * n, (i + 1) * n) of every XCD
            n = int(os.environ.get("UNET_LANE_CUS", "0"))
            self._lanes = [m.make_lane(x0.shape[2:], cu_range=(i * n, n) if n > 0 else None) for i in range(self.in_flight)]
        while len(self._gbufs) < len(mine):
            buf = m.flat_grads if not self._gbufs else torch.zeros_like(m.flat_grads)
            self._gbufs.append(buf)
            self._gptrs.append(m.grad_pointers(buf))
        main = tc.current_stream(m.device())
        for lane in self._lanes:
            lane["stream"].wait_stream(main)          # the parameters of the previous update, the caller's samples
        losses = []
        for k, b in enumerate(mine):
            lane = self._lanes[k % self.in_flight]
            with tc.stream(lane["stream"]):
                x, t = self.source(cur_data_index + b)
                if tuple(x.shape[2:]) != tuple(x0.shape[2:]):
                    raise ValueError("micro-steps in flight need samples of one size (lane plans were made for %s, got %s)"
                                     % (tuple(x0.shape[2:]), tuple(x.shape[2:])))
                # a lane's workspace holds its own filter packs: made by the lane's first micro-step of the step, reused by its later ones
                l = m.forward_backward_lane(lane, x, t, self._gptrs[k], p.cost_ce, p.cost_dice, p.cost_mse,
                                            packs_current=self.packs_reuse and k >= self.in_flight)
            l.record_stream(main)
            losses.append(l)
        for lane in self._lanes:
            main.wait_stream(lane["stream"])
        n = len(mine)
        E.check(E.lib.unet_sum_buffers(E.ptr_array([g.data_ptr() for g in self._gbufs[:n]]), n, m.flat_grads.data_ptr(),
                                       m.flat_grads.numel(), 1, main.cuda_stream))
        return losses

    def validate(self, test_in, test_out, output_model=None):
        """Thread D's body for one epoch (train.cpp:834-852, 890-895): eval() WITHOUT the running-statistics reset of
        prepare_for_inference (a bnorm model normalises with its running statistics here), forward, calc_losses on
        the full-resolution output only, mean over the test volumes -> [ce, dice, mse], appended to testing_errors."""
        m = output_model if output_model is not None else self.model
        was_training = m._training
        m.eval()
        acc = torch.zeros(3, dtype=torch.float64)
        with torch.no_grad():
            for x, t in zip(test_in, test_out):
                outs = m.forward(x)
                # unet_loss reports the full-resolution level's unweighted (ce, dice, mse) in slots 1..3: calc_losses(forward(x)[0], ...)
                losses, _ = m.loss(outs, t, True, True, True, 0, want_grad=False)
                acc += losses[1:4].double().cpu()
        m.train(was_training)
        errors = (acc / max(1, len(test_in))).tolist() if len(test_in) else []
        with self.model.error_mutex:
            for mm in {id(self.model): self.model, id(m): m}.values():
                mm.testing_errors.extend(errors)
        return errors

    def record_errors(self):
        """train.cpp:743-752: append the step's mean (ce, dice, mse) -- a host sync, call it sparingly"""
        e = (self._stats[1:4] / float(self.param.batch_size)).tolist()
        with self.model.error_mutex:
            self.model.training_errors.extend(e)
        return e
