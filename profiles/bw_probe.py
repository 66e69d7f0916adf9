#!/usr/bin/env python3
"""What this box's HBM delivers to plain streaming kernels (torch's own): the practical ceiling the HBM-side layers are priced
against next to the guide's 6.3 TB/s (MI355X_MICROARCH.md, HBM)."""
import torch

dev = "cuda:0"


def t(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for mb in (64, 134, 268, 536, 2048):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    y = torch.empty_like(x)
    ms_c = t(lambda: y.copy_(x))
    ms_s = t(lambda: x.sum())
    ms_f = t(lambda: y.fill_(1.0))
    print("%5d MB: copy %.3f ms = %.2f TB/s (r+w)   sum %.3f ms = %.2f TB/s (read)   fill %.3f ms = %.2f TB/s (write)"
          % (mb, ms_c, 2 * n * 4 / ms_c / 1e9, ms_s, n * 4 / ms_s / 1e9, ms_f, n * 4 / ms_f / 1e9), flush=True)
