#!/usr/bin/env python3
"""Host-side cost of one train step (enqueue only) vs the GPU time: tells when the step becomes launch-bound."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import unet_studio_amd as U
dev = "cuda:0"
n = 128
model = U.UNet3d(1, 6, U.default_feature(6), device=dev, dtype="bf16", seed=0)
param = U.TrainingParam(batch_size=1, epoch=10000, learning_rate=0.001)
src = U.SyntheticVolumes(1, 6, (n, n, n), dev, cache=4)
for i in range(4):
    src(i)
tr = U.Trainer(model, param, lambda i: src(i % 4), 0, 1)
for _ in range(10):
    tr.step()
torch.cuda.synchronize()
K = 30
t0 = time.perf_counter()
for _ in range(K):
    tr.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.3f ms/step, total %.3f ms/step" % ((t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3))
