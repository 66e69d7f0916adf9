#!/usr/bin/env python3
"""Micro-benchmark of single kernels through the C ABI (HIP events on torch's current stream), used for the
kernel experiments recorded in DESIGN.md.  Usage on the GPU box:  python profiles/bench_kernels.py [variant-env ...]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import unet_studio_amd as U  # noqa: E402

E = U.engine
DEV = "cuda:0"


def timeit(fn, iters=20):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def conv_case(cin, cout, n, stride=1, fused=True, what="fwd"):
    D = H = W = n
    od = (n - 1) // stride + 1
    x = torch.randn((D, H, W, cin), device=DEV).to(torch.bfloat16)
    w = torch.randn((cout, cin, 3, 3, 3), device=DEV) * 0.05
    b = torch.zeros(cout, device=DEV)
    y = torch.empty((od, od, od, cout), device=DEV, dtype=torch.bfloat16)
    dy = torch.randn((od, od, od, cout), device=DEV).to(torch.bfloat16)
    dx = torch.empty_like(x)
    dw = torch.zeros_like(w); db = torch.zeros_like(b)
    sc = torch.ones(cin, device=DEV); sh = torch.zeros(cin, device=DEV) + 0.1
    stats = torch.empty((cout, 2), device=DEV)
    nb = C.c_size_t()
    E.check(E.lib.unet_op_scratch_bytes(cin, cout, D, H, W, C.byref(nb)))
    scr = torch.empty(nb.value, dtype=torch.uint8, device=DEV)
    st = torch.cuda.current_stream(DEV).cuda_stream
    flops = 2.0 * cin * cout * 27 * od ** 3
    if what == "fwd":
        if fused:
            fn = lambda: E.check(E.lib.unet_op_conv3d_fwd_fused(1, 0, x.data_ptr(), sc.data_ptr(), sh.data_ptr(), 2, w.data_ptr(), b.data_ptr(),
                                                                y.data_ptr(), stats.data_ptr(), cin, cout, D, H, W, 3, stride, scr.data_ptr(), st))
        else:
            fn = lambda: E.check(E.lib.unet_op_conv3d_fwd(1, 0, x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), cin, cout, D, H, W, 3,
                                                          stride, scr.data_ptr(), st))
    elif what == "dgrad":
        fn = lambda: E.check(E.lib.unet_op_conv3d_bwd_data(1, 0, dy.data_ptr(), w.data_ptr(), dx.data_ptr(), cin, cout, D, H, W, 3, stride,
                                                           scr.data_ptr(), st))
    else:
        fn = lambda: E.check(E.lib.unet_op_conv3d_bwd_weight(1, 0, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), cin, cout, D, H, W,
                                                             3, stride, scr.data_ptr(), st))
    ms = timeit(fn)
    print("%-6s %3d->%3d @%3d^3 s%d fused=%d : %8.3f ms  %8.1f TFLOP/s  (%.1f%% of 2500)" %
          (what, cin, cout, n, stride, fused, ms, flops / ms / 1e9, flops / ms / 1e9 / 25.0), flush=True)
    return ms


if __name__ == "__main__":
    print("variant env:", {k: v for k, v in os.environ.items() if k.startswith("UNET_")})
    for what in ("fwd", "dgrad", "wgrad"):
        conv_case(32, 16, 128, what=what)
        conv_case(16, 16, 128, what=what)
        conv_case(64, 32, 64, what=what)
        conv_case(32, 32, 64, what=what)
        conv_case(128, 64, 32, what=what)
        conv_case(256, 128, 16, what=what)
        conv_case(256, 256, 8, what=what)
    conv_case(32, 16, 128, fused=False)
    conv_case(16, 32, 128, stride=2)
