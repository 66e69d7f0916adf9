#!/usr/bin/env python3
"""Launches only the weight gradient of one 3x3x3 layer (unet_op_conv3d_bwd_weight: wgrad kernel + slab reduce) a few times: the
target of the rocprofv3 --pmc passes of profiles/collect_counters.sh.   wgrad_kernel.py <cin> <cout> <size> [stride] [iters]"""
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
x = torch.randn((n, n, n, cin), device=dev).to(torch.bfloat16)
dy = torch.randn((no, no, no, cout), device=dev).to(torch.bfloat16)
dw = torch.zeros((cout, cin, 3, 3, 3), device=dev)
db = torch.zeros(cout, device=dev)
nb = C.c_size_t()
E.check(E.lib.unet_op_scratch_bytes(cin, cout, n, n, n, C.byref(nb)))
sc = torch.empty(nb.value, dtype=torch.uint8, device=dev)
for _ in range(iters):
    E.check(E.lib.unet_op_conv3d_bwd_weight(U.DTYPE_BF16, U.IMPL_AUTO, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), cin, cout,
                                            n, n, n, 3, stride, sc.data_ptr(), st))
torch.cuda.synchronize()
print("done")
