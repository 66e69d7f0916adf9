#!/usr/bin/env python3
"""One-GPU rehearsal of the collective calls bench.py / Trainer make at N > 1 (RCCL group of one rank): process-group init with
device_id, broadcast of the flat parameters, asynchronous all-reduces of the gradient buckets under the rest of the backward, of the loss statistics, between the
backward (whose parameter gradients come from the plan's side stream) and the fused optimizer step, barrier, destroy."""
import os, sys, time
import torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
import unet_studio_amd as U
n = 64
m = U.UNet3d(1, 6, U.default_feature(6), device="cuda:0", dtype="bf16", seed=0)
dist.broadcast(m.flat_params, 0)
m.create_optimizer(0.001)
src = U.SyntheticVolumes(1, 6, (n, n, n), "cuda:0", cache=2)
ref = U.UNet3d(1, 6, U.default_feature(6), device="cuda:0", dtype="bf16", seed=0)
ref.create_optimizer(0.001)
stats = torch.zeros(4, device="cuda:0")
for step in range(5):
    x, t = src(step % 2)
    works = []
    l = m.forward_backward_bucketed(x, t, lambda lo, hi: works.append(dist.all_reduce(m.flat_grads[lo:hi], op=dist.ReduceOp.SUM, async_op=True)))
    stats.copy_(l)
    dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    for w in works:
        w.wait()
    m.optimizer.step(grad_scale=1.0, clip_norm=12.0)
    ref.forward_backward(x, t)
    ref.optimizer.step(grad_scale=1.0, clip_norm=12.0)
dist.barrier()
torch.cuda.synchronize()
same = torch.equal(m.flat_params, ref.flat_params)
print("rccl single-rank rehearsal: params identical to the no-collective run:", same, "loss", float(stats[0]))
dist.destroy_process_group()
sys.exit(0 if same else 1)
