#!/usr/bin/env python3
"""Dominant conv kernel (32->16 @128^3) with and without the statistics epilogue, boost vs sustained clocks."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import unet_studio_amd as U
E = U.engine
dev = "cuda:0"
cin, cout, n = 32, 16, 128
x = torch.randn((n, n, n, cin), device=dev).to(torch.bfloat16)
w = torch.randn((cout, cin, 3, 3, 3), device=dev) * 0.05
b = torch.zeros(cout, device=dev)
y = torch.empty((n, n, n, cout), device=dev, dtype=torch.bfloat16)
nb = C.c_size_t(); E.check(E.lib.unet_op_scratch_bytes(cin, cout, n, n, n, C.byref(nb)))
sc = torch.empty(nb.value, dtype=torch.uint8, device=dev); wp = torch.empty(nb.value, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
E.check(E.lib.unet_op_conv3d_pack(1, w.data_ptr(), wp.data_ptr(), cin, cout, n, n, n, 3, 1, st))
def run(stats):
    E.check(E.lib.unet_op_conv3d_fwd_packed(1, x.data_ptr(), wp.data_ptr(), b.data_ptr(), y.data_ptr(), sc.data_ptr() if stats else None,
                                            cin, cout, n, n, n, 3, 1, st))
def t(stats, iters):
    for _ in range(3): run(stats)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run(stats)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for stats in (True, False):
    for iters in (5, 20, 100):
        ms = t(stats, iters)
        print("stats=%d iters=%3d : %.4f ms  %.0f TFLOP/s" % (stats, iters, ms, 2.0 * cin * cout * 27 * n ** 3 / ms / 1e9))
