#!/usr/bin/env python3
"""Soak / sanity run: many optimizer steps of the benchmarked configuration on a fixed set of samples; the loss must go down and
stay finite (catches races, leaks and optimizer-path mistakes that single-step parity tests cannot see)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import unet_studio_amd as U
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = "cuda:0"
m = U.UNet3d(1, 6, U.default_feature(6), device=dev, dtype="bf16", seed=0)
src = U.SyntheticVolumes(1, 6, (n, n, n), dev, cache=2)
tr = U.Trainer(m, U.TrainingParam(batch_size=1, epoch=steps * 2, learning_rate=0.01), lambda i: src(i % 2), 0, 1)
hist = []
t0 = time.time()
for i in range(steps):
    s = tr.step()
    if i % 25 == 0 or i == steps - 1:
        hist.append(float(s[0]))
        print("step %4d loss %.4f  grad-norm %.3f  mem %.2f GB" % (i, hist[-1], float(m.optimizer.last_grad_norm), torch.cuda.memory_allocated() / 2**30), flush=True)
torch.cuda.synchronize()
print("%.1f ms/step over %d steps; loss %.4f -> %.4f" % ((time.time() - t0) / steps * 1e3, steps, hist[0], hist[-1]))
ok = all(map(lambda v: v == v and v < 1e4, hist)) and hist[-1] < hist[0]
sys.exit(0 if ok else 1)
