#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): voxels/s of a 3D U-Net train step (forward + 5-level CE/Dice/MSE loss +
backward + /batch_size + clip_grad_norm_(12) + SGD-Nesterov) on synthetic 128^3 single-channel volumes, bf16
activations / fp32 master weights, default architecture (train.cpp:1054-1069), in=1, out=6.

  python bench.py --gpus N --steps K --warmup W

N > 1: either launched by torch.distributed.run (one rank per GPU: RANK / LOCAL_RANK / WORLD_SIZE in the environment), or --
with WORLD_SIZE unset -- this script starts its N ranks itself as fresh child processes (before anything touches the GPU in
the parent), the way train_unet::start spawns one worker per visible device (train.cpp:581-606,962-971).

One "step" = one optimizer step with batch_size = N (one 128^3 sample per GPU per step, weak scaling): the flat fp32
gradient buffer is summed over ranks with RCCL all-reduces (in buckets, under the backward), then every rank applies the
identical update.  Rank 0 prints ONE JSON line:
  * `roofline`        the launch with the largest share of the step's conv kernel time (found live: every op of three profiled
                      steps is bracketed by HIP events on its launch stream, unet_profile_begin/end), algorithmic FLOPs / its time;
  * `conv_mfma_frac`  conv forward + dgrad + wgrad FLOPs of the step / the summed time of those kernels / the bf16 MFMA peak
                      (north_star: >= 40 % on conv3d fwd+bwd), next to `step_mfma_frac` (same FLOPs / the whole step's wall time);
  * `roofline_kernels` micro-benchmarks through the single-op C ABI of the heaviest forward conv and the heaviest weight gradient;
  * `cpu_baseline`    the ATen-CPU executor of oracle/aten_ref.py (the reference's CPU path = libtorch CPU kernels in unet.cpp
                      order) on a bounded sample, rank 0, N=1 only.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16 = 2.5e15   # dense bf16 MFMA, MI355X_MICROARCH.md
PEAK_F32_MATRIX = 157.3e12
PEAK_HBM = 8.0e12    # spec; ~6.3e12 achievable (MI355X_MICROARCH.md, HBM)


def _time_launches(run, iters):
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def dominant_kernel(U, size, dtype_name, iters=20):
    """Times the heaviest forward conv3d of the step (decode0.0: 32->16 @ size^3, 3x3x3) as launched through the C ABI.
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
    sec = _time_launches(run, iters)
    flops = 2.0 * cin * cout * 27 * D * H * W
    return flops, sec


os.environ.setdefault("UNET_OP_POLITE", "1")   # op-level weight-gradient launches in the configuration the train step uses (engine.cpp)


def dominant_wgrad(U, size, dtype_name, cin=32, cout=16, iters=20):
    """The heaviest weight gradient of the step (decode0.0: dW of conv3d cin->cout 3x3x3 @ size^3) through
    unet_op_conv3d_bwd_weight: the wgrad kernel + its slab reduce, as a plan launches them."""
    import ctypes as C
    E = U.engine
    D = H = W = size
    dt = torch.bfloat16 if dtype_name == "bf16" else torch.float32
    edt = U.DTYPE_BF16 if dtype_name == "bf16" else U.DTYPE_F32
    dev = torch.device("cuda", torch.cuda.current_device())
    x = torch.randn((D, H, W, cin), device=dev).to(dt)
    dy = torch.randn((D, H, W, cout), device=dev).to(dt)
    dw = torch.zeros((cout, cin, 3, 3, 3), device=dev)
    db = torch.zeros(cout, device=dev)
    nb = C.c_size_t()
    E.check(E.lib.unet_op_scratch_bytes(cin, cout, D, H, W, C.byref(nb)))
    sc = torch.empty(nb.value, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream

    def run():
        E.check(E.lib.unet_op_conv3d_bwd_weight(edt, U.IMPL_AUTO, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), cin, cout,
                                                D, H, W, 3, 1, sc.data_ptr(), st))
    sec = _time_launches(run, iters)
    return 2.0 * cin * cout * 27 * D * H * W, sec


def recorded_traffic(name):
    """HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE),
    RECORDED by profiles/collect_traffic.sh into profiles/<name>.json -- not a measurement of this run; returned with its source."""
    path = os.path.join("profiles", name + ".json")
    try:
        with open(os.path.join(ROOT, path)) as f:
            d = json.load(f)
        return d.get("hbm_bytes_per_launch"), "recorded: %s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE; %s)" % (path, d.get("kernel", "?"))
    except Exception:
        return None, None


def cpu_baseline(size, budget_steps=3):   # ~3 s per 128^3 step on 16 cores: 1 warm-up + 3 timed steps = ~12 s of CPU work
    """ATen-CPU train micro-step (forward + losses + backward + step epilogue), fp32, all host cores."""
    from oracle import aten_ref as A
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # threads = the process's CPU share: a one-GPU box of this pool grants 16 cores to a job (worker pools must be sized to it) whatever
    # the host's core count; torch threads beyond the share only oversubscribe it
    cores = max(1, min(avail, 16))
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
                      "(oracle/aten_ref.py), torch %s; %d threads = the job's CPU share on a one-GPU box (16 cores; %d visible to the process)"
                      % (budget_steps, size, torch.__version__, cores, avail),
            "ms_per_step": dt * 1e3}


def op_flops(o):
    """algorithmic 2*MAC of one conv / conv_trans op (forward = dgrad = wgrad)"""
    if o["kind"] == 1:
        v = o["out_dims"][0] * o["out_dims"][1] * o["out_dims"][2]
        return 2.0 * o["ks"] ** 3 * o["cin"] * o["cout"] * v
    if o["kind"] == 2:
        v = o["in_dims"][0] * o["in_dims"][1] * o["in_dims"][2]
        return 2.0 * 8 * o["cin"] * o["cout"] * v
    return 0.0


def op_bytes(o, cat, esize):
    """algorithmic HBM bytes of one conv / conv_trans launch: every operand once (activations at `esize` bytes per element, the packed
    filter at esize, the weight gradient in fp32); re-reads (halos, accumulating epilogues, slabs) are NOT algorithmic"""
    vi = o["in_dims"][0] * o["in_dims"][1] * o["in_dims"][2]
    vo = o["out_dims"][0] * o["out_dims"][1] * o["out_dims"][2]
    taps = o["ks"] ** 3 if o["kind"] == 1 else 8
    act = (vi * o["cin"] + vo * o["cout"]) * esize
    return act + taps * o["cin"] * o["cout"] * (4 if cat == "wgrad" else esize)


def profile_steps(U, trainer, plan, peak, nsteps=3):
    """Per-op HIP-event brackets (include/unet_hip.h: unet_profile_begin/end) over `nsteps` optimizer steps: kernel time by family,
    conv MFMA fraction, and the single launch with the largest share of the conv time."""
    E = U.engine
    trainer.step()
    torch.cuda.synchronize()
    with E.profile() as pr:
        for _ in range(nsteps):
            trainer.step()
        torch.cuda.synchronize()
    ops = plan.ops()
    fam, per = {}, {}
    for op, cat, ms in pr.records:
        fam[cat] = fam.get(cat, 0.0) + ms / nsteps
        per[(op, cat)] = per.get((op, cat), 0.0) + ms / nsteps
    conv_ms = sum(fam.get(c, 0.0) for c in ("conv_fwd", "dgrad", "wgrad"))
    flops = plan.flops_fwd + plan.flops_bwd
    best = None
    for (op, cat), ms in per.items():
        if cat in ("conv_fwd", "dgrad", "wgrad") and op >= 0 and (best is None or ms > best[2]):
            best = (op, cat, ms)
    out = {"per_op": per, "kernel_ms_by_family": {k: round(v, 4) for k, v in sorted(fam.items())},
           "conv_kernel_ms": conv_ms, "conv_mfma_frac": flops / (conv_ms * 1e-3) / peak if conv_ms > 0 else None,
           "profiled_steps": nsteps, "launch_brackets_per_step": len(pr.records) // nsteps}
    if best is not None:
        o = ops[best[0]]
        fl = op_flops(o)
        out["dominant"] = {"op": o["name"], "pass": best[1], "flops": fl, "ms": best[2], "bytes": op_bytes(o, best[1], 2 if peak == PEAK_BF16 else 4),
                           "share_of_conv_time": best[2] / conv_ms if conv_ms > 0 else None,
                           "shape": "%d->%d k%d s%d @%s" % (o["cin"], o["cout"], o["ks"], o["stride"], "x".join(str(d) for d in o["out_dims"]))}
    return out


STEP_TRAFFIC_FILE = "profiles/step_hbm_traffic.json"   # written by profiles/collect_step_traffic.sh (rocprofv3 --pmc passes over whole steps)


def conv_group(o):
    """the three groups the round-2 review separated: stride-1 convs at >= 32^3 (95 % of the FLOPs), conv_trans + stride-2 convs,
    stride-1 convs at <= 16^3; the 1x1x1 heads apart"""
    if o["kind"] == 2 or o["stride"] == 2:
        return "conv_trans+stride2"
    if o["ks"] == 1:
        return "heads"
    v = o["out_dims"][0] * o["out_dims"][1] * o["out_dims"][2]
    return "stride1_ge_32^3" if v >= 32 ** 3 else "stride1_le_16^3"


def step_roofline(plan, prof_per, ms_per_step, peak, esize, n_params):
    """`roofline_step`: the whole step against the HBM roofline, and the conv time by group (live per-op profile).
    algorithmic bytes = every conv / conv_trans operand once per pass (forward, dgrad, wgrad: op_bytes) + the optimizer's six passes
    over the fp32 parameters (p, g, m read; p, m, g written); norm / activation / loss passes are NOT algorithmic (ideal fusion folds
    them into the producing kernels).  counter_bytes = the PMC total recorded by profiles/collect_step_traffic.sh, when present."""
    ops = plan.ops()
    alg = 0.0
    first = True
    for o in ops:
        if o["kind"] not in (1, 2):
            continue
        alg += op_bytes(o, "conv_fwd", esize) + op_bytes(o, "wgrad", esize)
        if not first:
            alg += op_bytes(o, "dgrad", esize)   # the first conv's input needs no gradient
        first = False
    alg += 6.0 * 4 * n_params
    groups = {}
    for (op, cat), ms in prof_per.items():
        if op < 0 or cat not in ("conv_fwd", "dgrad", "wgrad"):
            continue
        g = groups.setdefault(conv_group(ops[op]), {"ms": 0.0, "flops": 0.0})
        g["ms"] += ms
        g["flops"] += op_flops(ops[op])
    tot_fl = sum(g["flops"] for g in groups.values()) or 1.0
    out = {"algorithmic_bytes": alg, "hbm_frac": alg / (ms_per_step * 1e-3) / PEAK_HBM, "counter_bytes": None, "counter_source": None,
           "conv_groups": {k: {"ms_per_step": round(g["ms"], 4), "flop_share": round(g["flops"] / tot_fl, 4),
                               "mfma_frac": round(g["flops"] / (g["ms"] * 1e-3) / peak, 4) if g["ms"] > 0 else None}
                           for k, g in sorted(groups.items())}}
    try:
        with open(os.path.join(ROOT, STEP_TRAFFIC_FILE)) as f:
            d = json.load(f)
        out["counter_bytes"] = d.get("hbm_bytes_per_step")
        out["counter_source"] = "recorded: %s (%s)" % (STEP_TRAFFIC_FILE, d.get("note", "rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE over whole steps"))
    except Exception:
        pass
    return out


def cpp_host_step(n, steps, warmup, dtype):
    """the same optimizer step driven by the C++ drop-in host (include/unet.hpp + unet_host.cpp: loss_and_backward + sgd_step --
    what train.cpp's thread C would call), timed by unet-studio_amd/bench_host in a child process of its own"""
    import subprocess
    exe = os.path.join(ROOT, "unet-studio_amd", "bench_host")
    if not os.path.exists(exe):
        return {"error": "unet-studio_amd/bench_host is not built (__graft_entry__.build())"}
    try:
        r = subprocess.run([exe, str(n), str(steps), str(warmup), dtype], capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            return {"error": "bench_host rc %d: %s" % (r.returncode, (r.stderr or r.stdout)[-300:])}
        return json.loads(line[-1])
    except Exception as e:   # noqa: BLE001  (a reported field, never fatal for the bench line)
        return {"error": str(e)}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n):
    """WORLD_SIZE unset and --gpus n > 1: be the launcher.  Nothing in this process has touched the GPU (device_count() does not
    initialise it), the ranks are fresh children; rank 0 inherits stdout and prints the one JSON line."""
    have = torch.cuda.device_count()
    if have < n:
        raise SystemExit("bench.py --gpus %d: %d devices needed, %d visible -- one rank per GPU (train.cpp:962-971 builds one replica "
                         "per visible device), ranks cannot share a device under RCCL" % (n, n, have))
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    for q in pending:      # a failed rank leaves the others stuck in a collective: end exactly our children
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    if rc:
        raise SystemExit("bench.py: a rank exited with code %d" % rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--step-times", action="store_true", help="diagnostic: event-time every step of the timed region, print the slowest to stderr")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-op HIP-event profile (roofline / conv_mfma_frac)")
    ap.add_argument("--no-kernels", action="store_true", help="skip the single-kernel micro-benchmarks (roofline_kernels)")
    ap.add_argument("--no-cpp-host", action="store_true", help="skip timing the same step through the C++ host (unet-studio_amd/bench_host)")
    ap.add_argument("--batch", type=int, default=None, help="(default: 8 on one GPU, 0 = off on several) the reference's step shape (train.cpp:604-606: batch_size micro-steps per update, "
                    "train.hpp:12 default 32; SURVEY 8(d) config 3 asks for 8): B micro-steps per GPU per optimizer step, timed AFTER the "
                    "headline and reported as `batch<B>` inside the same line; 0 = skip")
    # BASELINE.json configs[4] (not the headline line): --size 256 --in-channels 2 --augment --no-cpu-baseline
    ap.add_argument("--in-channels", type=int, default=1)
    ap.add_argument("--augment", action="store_true", help="augment every sample on the GPU inside the timed step (unet_augment_run)")
    ap.add_argument("--prefetch", action="store_true", help="with --augment: sample ring, sample i+1 is augmented on a side stream during step i")
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        spawn_ranks(a.gpus)
        return
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.batch is None:     # SURVEY 8(d) config 3 asks for batch_size 8 on ONE GPU; the scaling runs time the headline step only
        a.batch = 8 if world == 1 else 0
    if a.gpus != world:
        raise SystemExit("--gpus %d does not match WORLD_SIZE=%d" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if torch.cuda.device_count() <= local:
        raise SystemExit("bench.py --gpus %d: %d devices needed, %d visible" % (a.gpus, a.gpus, torch.cuda.device_count()))
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
    param = U.TrainingParam(batch_size=world, epoch=max(10000, a.steps + a.warmup + 16), learning_rate=0.001)
    src = U.SyntheticVolumes(cin, 6, (n, n, n), dev, cache=4)   # samples resident in HBM before the timed region
    for i in range(world * 2):
        src(i % 4)
    feed = lambda i: src(i % 4)   # noqa: E731
    if a.augment:   # the resident template sample is augmented anew (seed = step-unique sample index) inside every timed step
        feed = U.AugmentedVolumes(lambda i: src(i % 4))
        if a.prefetch and hasattr(U, "PrefetchedVolumes"):
            feed = U.PrefetchedVolumes(feed, stride=world)
    trainer = U.Trainer(model, param, feed, rank, world)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the interpreter's cyclic garbage collector stays out of the timed region (a full collection of a torch process's heap is a
    # multi-millisecond host pause; the step allocates no cycles): collected before the warm-up -- not between warm-up and timed steps,
    # where the ~0.1 s it takes leaves the GPU idle and the first timed steps run 0.1-0.3 ms long while it clocks up again -- re-enabled after
    import gc
    gc.collect()
    gc.disable()
    for _ in range(a.warmup):
        trainer.step()
    sync()
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)] if a.step_times else None
    t0 = time.perf_counter()
    if step_ev:
        step_ev[0].record()
    for k in range(a.steps):
        trainer.step()
        if step_ev:
            step_ev[k + 1].record()
    sync()
    dt = time.perf_counter() - t0
    gc.enable()
    if step_ev:   # diagnostic: where inside the timed region the time went (GPU-side interval between consecutive steps' last kernels)
        iv = [step_ev[k].elapsed_time(step_ev[k + 1]) for k in range(a.steps)]
        order = sorted(range(a.steps), key=lambda k: -iv[k])[:3]
        sys.stderr.write("step intervals (ms): median %.3f; slowest %s\n" % (sorted(iv)[len(iv) // 2], ", ".join("#%d %.3f" % (k, iv[k]) for k in order)))
    tt = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt)
    # who took part: lets a reader of the line confirm the rank count and that the collectives ran over RCCL (gathered FROM the ranks)
    me = {"rank": rank, "device": torch.cuda.get_device_name(dev), "local_device": str(dev), "pid_host": os.uname().nodename}
    comm_info = {"backend": None, "world_size": 1, "rccl_version": None, "ranks": [me]}
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, me)
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            ver = None
        comm_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rccl_version": ver, "ranks": gathered,
                     "gradient_allreduce": "flat fp32 gradient buffer, %d elements, bucketed under the backward (train.py: Trainer.step)" % model.flat_grads.numel()}
    loss = float(trainer._stats[0]) / max(1, len(range(rank, world, world)))

    # the reference's own step shape: batch_size micro-steps per update (train.cpp:604-606,759-761), here B per GPU per step.  Clip,
    # SGD, the filter pack and the all-reduce are paid once per B samples.  Every rank takes part (collectives inside).
    batch_line = None
    if a.batch > 0:
        pb = U.TrainingParam(batch_size=world * a.batch, epoch=param.epoch, learning_rate=0.001)
        tb = U.Trainer(model, pb, feed, rank, world)
        bsteps = max(2, min(a.steps, 48 // a.batch))
        tb.step()
        sync()
        t0 = time.perf_counter()
        for _ in range(bsteps):
            tb.step()
        sync()
        dtb = time.perf_counter() - t0
        tb_ = torch.tensor([dtb], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(tb_, op=dist.ReduceOp.MAX)
        dtb = float(tb_)
        batch_line = {"micro_steps_per_gpu_per_update": a.batch, "global_batch": world * a.batch, "steps": bsteps,
                      "ms_per_step": dtb / bsteps * 1e3, "ms_per_sample": dtb / bsteps / a.batch * 1e3,
                      "value": float(n) ** 3 * world * a.batch * bsteps / dtb, "unit": "voxels/s",
                      "in_flight": int(os.environ.get("UNET_MICRO_IN_FLIGHT", "1")) if hasattr(tb, "in_flight") else 1}

    if rank == 0:
        plan = model.plan_for((n, n, n))
        vox = float(n) ** 3 * world * a.steps
        value = vox / dt
        step_flops = plan.flops_fwd + plan.flops_bwd
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
            "last_loss": loss,
        }
        out["comm"] = comm_info
        kernels = []
        if a.batch > 0:
            out["batch%d" % a.batch] = batch_line
        if world == 1 and not a.no_cpp_host and not a.augment and cin == 1:
            out["cpp_host"] = cpp_host_step(n, a.steps, a.warmup, a.dtype)
        if world == 1 and not a.no_profile:
            # (ranks > 1 would need the other ranks to join the profiled steps' collectives: the profile is a single-GPU measurement)
            prof = profile_steps(U, trainer, plan, peak)
            out["conv_mfma_frac"] = prof["conv_mfma_frac"]
            out["conv_kernel_ms_per_step"] = prof["conv_kernel_ms"]
            out["kernel_ms_by_family"] = prof["kernel_ms_by_family"]
            out["roofline_step"] = step_roofline(plan, prof["per_op"], out["ms_per_step"], peak, 2 if a.dtype == "bf16" else 4,
                                                 int(model.flat_params.numel()))
            d = prof.get("dominant")
            if d:
                # the roofline that binds this launch: the longer of (algorithmic FLOPs / dense MFMA peak) and (algorithmic bytes / HBM peak)
                t_mfma, t_hbm, sec = d["flops"] / peak, d["bytes"] / PEAK_HBM, d["ms"] * 1e-3
                tr_name = {("wgrad", "32->16 k3 s1 @128x128x128"): "wgrad_kernel_traffic",
                           ("conv_fwd", "32->16 k3 s1 @128x128x128"): "dominant_kernel_traffic"}.get((d["pass"], d["shape"])) if a.dtype == "bf16" else None
                tr_b, tr_src = recorded_traffic(tr_name) if tr_name else (None, None)
                rl = {"bound": "hbm", "achieved": d["bytes"] / sec / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s", "frac": t_hbm / sec} if t_hbm >= t_mfma else \
                     {"bound": "mfma", "achieved": d["flops"] / sec / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s", "frac": t_mfma / sec}
                rl.update({"traffic": tr_b, "traffic_source": tr_src,
                           "kernel": "%s of %s (%s): the launch with the largest share of the step's conv kernel time (%.1f %%), "
                                     "HIP events around it inside %d profiled steps"
                                     % (d["pass"], d["op"], d["shape"], 100.0 * d["share_of_conv_time"], prof["profiled_steps"]),
                           "avg_launch_ms": d["ms"], "algorithmic_bytes": d["bytes"], "algorithmic_flops": d["flops"],
                           "mfma_frac": t_mfma / sec, "hbm_frac": t_hbm / sec})
                out["roofline"] = rl
        if n == 128 and cin == 1 and not a.no_kernels:
            esz = 2 if a.dtype == "bf16" else 4

            def entry(name, fl, by, sec, tr_name):
                t_mfma, t_hbm = fl / peak, by / PEAK_HBM
                tr_b, tr_src = recorded_traffic(tr_name) if a.dtype == "bf16" else (None, None)
                e = {"kernel": name, "avg_launch_ms": sec * 1e3, "algorithmic_flops": fl, "algorithmic_bytes": by,
                     "mfma_frac": t_mfma / sec, "hbm_frac": t_hbm / sec, "traffic": tr_b, "traffic_source": tr_src}
                e.update({"bound": "hbm", "achieved": by / sec / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s", "frac": t_hbm / sec} if t_hbm >= t_mfma else
                         {"bound": "mfma", "achieved": fl / sec / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s", "frac": t_mfma / sec})
                return e
            kflops, ksec = dominant_kernel(U, n, a.dtype)
            kernels.append(entry("conv3d fwd 32->16 3x3x3 @%d^3 (decode0.0, 23.7%% of forward FLOPs), unet_op_conv3d_fwd_packed" % n, kflops,
                                 n ** 3 * (32 + 16) * esz + 27 * 32 * 16 * esz, ksec, "dominant_kernel_traffic"))
            wflops, wsec = dominant_wgrad(U, n, a.dtype)
            kernels.append(entry("conv3d wgrad 32->16 3x3x3 @%d^3 (decode0.0) + slab reduce, unet_op_conv3d_bwd_weight" % n, wflops,
                                 n ** 3 * (32 + 16) * esz + 27 * 32 * 16 * 4, wsec, "wgrad_kernel_traffic"))
            out["roofline_kernels"] = kernels
            if "roofline" not in out:
                out["roofline"] = dict(kernels[0])
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
