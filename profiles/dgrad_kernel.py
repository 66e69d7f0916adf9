#!/usr/bin/env python3
"""Launches only the input gradient of one 3x3x3 layer (unet_op_conv3d_bwd_data) a few times: target of profiles/collect_counters.sh.
   dgrad_kernel.py <cin> <cout> <size> [stride] [iters]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import unet_studio_amd as U  # noqa: E402

E = U.engine
cin, cout, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
stride = int(sys.argv[4]) if len(sys.argv) > 4 else 1
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream
no = n // stride
dy = torch.randn((no, no, no, cout), device=dev).to(torch.bfloat16)
w = torch.randn((cout, cin, 3, 3, 3), device=dev) * 0.05
dx = torch.empty((n, n, n, cin), device=dev, dtype=torch.bfloat16)
nb = C.c_size_t()
E.check(E.lib.unet_op_scratch_bytes(cin, cout, n, n, n, C.byref(nb)))
sc = torch.empty(nb.value, dtype=torch.uint8, device=dev)
for _ in range(3):
    E.check(E.lib.unet_op_conv3d_bwd_data(U.DTYPE_BF16, U.IMPL_AUTO, dy.data_ptr(), w.data_ptr(), dx.data_ptr(), cin, cout, n, n, n, 3, stride, sc.data_ptr(), st))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    E.check(E.lib.unet_op_conv3d_bwd_data(U.DTYPE_BF16, U.IMPL_AUTO, dy.data_ptr(), w.data_ptr(), dx.data_ptr(), cin, cout, n, n, n, 3, stride, sc.data_ptr(), st))
e1.record()
torch.cuda.synchronize()
print("dgrad %d<-%d @%d^3 s%d: %.4f ms per call (incl. filter pack)" % (cin, cout, n, stride, e0.elapsed_time(e1) / iters))
