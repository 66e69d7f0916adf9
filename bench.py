#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): voxels/s of a 3D U-Net train step (forward + 5-level CE/Dice/MSE loss +
backward + /batch_size + clip_grad_norm_(12) + SGD-Nesterov) on synthetic 128^3 single-channel volumes, bf16
activations / fp32 master weights, default architecture (train.cpp:1054-1069), in=1, out=6.

  python bench.py --gpus N --steps K --warmup W      (N>1: launched by torch.distributed.run, one rank per GPU)

One "step" = one optimizer step with batch_size = N (one 128^3 sample per GPU per step, weak scaling): the flat fp32
gradient buffer is summed over ranks with one RCCL all-reduce, then every rank applies the identical update.
Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the heaviest conv3d of the step), timed live
with HIP events on the launch stream; `cpu_baseline` is the ATen-CPU executor of oracle/aten_ref.py (the reference's
CPU path = libtorch CPU kernels in unet.cpp order) on a bounded sample, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16 = 2.5e15   # dense bf16 MFMA, MI355X_MICROARCH.md
PEAK_F32_MATRIX = 157.3e12
PEAK_HBM = 8.0e12


def dominant_kernel(U, size, dtype_name, iters=20):
    """Times the heaviest conv3d of the step (decode0.0: 32->16 @ size^3, 3x3x3) as launched through the C ABI.
    Returns (algorithmic flops per launch, avg seconds per launch)."""
    import ctypes as C
    E = U.engine
    cin, cout = 32, 16
    D = H = W = size
    dt = torch.bfloat16 if dtype_name == "bf16" else torch.float32
    edt = U.DTYPE_BF16 if dtype_name == "bf16" else U.DTYPE_F32
    dev = torch.device("cuda", torch.cuda.current_device())
    x = torch.randn((D, H, W, cin), device=dev).to(dt)
    w = torch.randn((cout, cin, 3, 3, 3), device=dev) * 0.05
    b = torch.zeros(cout, device=dev)
    y = torch.empty((D, H, W, cout), device=dev, dtype=dt)
    nb = C.c_size_t()
    E.check(E.lib.unet_op_scratch_bytes(cin, cout, D, H, W, C.byref(nb)))
    sc = torch.empty(nb.value, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream

    if dtype_name == "bf16":
        wp = torch.empty(nb.value, dtype=torch.uint8, device=dev)
        # exactly what a plan does for decode0.0: filters packed once per step, then ONE launch of the conv kernel per sample
        # (plain activated input, bias + bf16 store + norm-statistics partials in the epilogue)
        E.check(E.lib.unet_op_conv3d_pack(edt, w.data_ptr(), wp.data_ptr(), cin, cout, D, H, W, 3, 1, st))

        def run():
            E.check(E.lib.unet_op_conv3d_fwd_packed(edt, x.data_ptr(), wp.data_ptr(), b.data_ptr(), y.data_ptr(), sc.data_ptr(),
                                                    cin, cout, D, H, W, 3, 1, st))
    else:   # fp32 engine: the fp32 matrix-core conv (k_conv_f32_mfma) behind the operator entry point (+ its ~5 us filter repack)
        def run():
            E.check(E.lib.unet_op_conv3d_fwd(edt, U.IMPL_AUTO, x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), cin, cout, D, H, W,
                                             3, 1, sc.data_ptr(), st))
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) * 1e-3 / iters
    flops = 2.0 * cin * cout * 27 * D * H * W
    return flops, sec


def measured_traffic():
    """HBM bytes per launch of the dominant kernel from rocprofv3 PMC passes (FETCH_SIZE doubled as MI355X_MICROARCH.md
    prescribes for gfx950, + WRITE_SIZE), recorded by profiles/collect_traffic.sh into profiles/dominant_kernel_traffic.json."""
    try:
        with open(os.path.join(ROOT, "profiles", "dominant_kernel_traffic.json")) as f:
            return json.load(f).get("hbm_bytes_per_launch")
    except Exception:
        return None


def cpu_baseline(size, budget_steps=3):   # ~3 s per 128^3 step on 16 cores: 1 warm-up + 3 timed steps = ~12 s of CPU work
    """ATen-CPU train micro-step (forward + losses + backward + step epilogue), fp32, all host cores."""
    from oracle import aten_ref as A
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))   # a 1-GPU box's CPU share is 16 cores
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    m = A.UNet3dRef(1, 6, A.default_feature(6))
    m.train()
    opt = m.create_optimizer(0.001)
    x, t = A.synthetic_sample(1, 6, (size, size, size), 0)

    def step():
        outs = m(x)
        loss, _ = A.deep_supervision_loss(outs, t, 6)
        loss.backward()
        A.train_step_epilogue(m, opt, 1)
    step()  # warm-up
    t0 = time.time()
    for _ in range(budget_steps):
        step()
    dt = (time.time() - t0) / budget_steps
    return {"value": size ** 3 / dt, "unit": "voxels/s", "cores": cores, "kind": "port",
            "sample": "%d train steps (1 warm-up) of the default arch at %d^3, fp32, ATen CPU kernels in unet.cpp order "
                      "(oracle/aten_ref.py), torch %s" % (budget_steps, size, torch.__version__),
            "ms_per_step": dt * 1e3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    # BASELINE.json configs[4] (not the headline line): --size 256 --in-channels 2 --augment --no-cpu-baseline
    ap.add_argument("--in-channels", type=int, default=1)
    ap.add_argument("--augment", action="store_true", help="augment every sample on the GPU inside the timed step (unet_augment_run)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (a.gpus, a.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = "cuda:%d" % local
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device(dev))

    import unet_studio_amd as U
    n = a.size
    cin = a.in_channels
    model = U.UNet3d(cin, 6, U.default_feature(6), device=dev, dtype=a.dtype, seed=0)
    if world > 1:  # same initial weights everywhere (the reference broadcasts every step: train.cpp:573-579)
        dist.broadcast(model.flat_params, 0)
    param = U.TrainingParam(batch_size=world, epoch=max(10000, a.steps + a.warmup + 1), learning_rate=0.001)
    src = U.SyntheticVolumes(cin, 6, (n, n, n), dev, cache=4)   # samples resident in HBM before the timed region
    for i in range(world * 2):
        src(i % 4)
    feed = src
    if a.augment:   # the resident template sample is augmented anew (seed = step-unique sample index) inside every timed step
        aug = U.AugmentedVolumes(lambda i: src(i % 4))
        feed = aug
    trainer = U.Trainer(model, param, (lambda i: feed(i)) if a.augment else (lambda i: src(i % 4)), rank, world)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        trainer.step()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        trainer.step()
    sync()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt)
    loss = float(trainer._stats[0]) / max(1, len(range(rank, world, world)))

    if rank == 0:
        plan = model.plan_for((n, n, n))
        vox = float(n) ** 3 * world * a.steps
        value = vox / dt
        step_flops = plan.flops_fwd + plan.flops_bwd
        kflops, ksec = dominant_kernel(U, n, a.dtype)
        peak = PEAK_BF16 if a.dtype == "bf16" else PEAK_F32_MATRIX
        out = {
            "metric": "voxels/sec 3D U-Net train step @%d^3 %s" % (n, a.dtype), "value": value, "unit": "voxels/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": "%s: train step (fwd+loss+bwd+clip+SGD-Nesterov), default UNet3d arch "
                                   "(train.cpp:1054-1069), in=%d out=6, %d^3 volumes, 1 sample per GPU per step%s"
                                   % ("configs[2]" if (cin, n, a.augment, a.dtype) == (1, 128, False, "bf16") else "configs[4]-style" if a.augment else
                                      "variant", cin, n, ", on-GPU visual_perception_augmentation inside the step" if a.augment else ""),
                       "global_batch": world, "volume": [n, n, n], "parallelism": "dp%d" % world,
                       "flops_per_step_per_sample": step_flops, "params": int(model.flat_params.numel())},
            "step_mfma_frac": (step_flops * world * a.steps / dt) / (peak * world),
            "roofline": {"bound": "mfma", "achieved": kflops / ksec / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s",
                         "frac": kflops / ksec / peak, "traffic": measured_traffic() if a.dtype == "bf16" else None,
                         "kernel": "conv3d fwd 32->16 3x3x3 @%d^3 (decode0.0, 23.7%% of forward FLOPs)" % n,
                         "avg_launch_ms": ksec * 1e3},
            "last_loss": loss,
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
