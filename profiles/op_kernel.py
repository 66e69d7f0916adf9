#!/usr/bin/env python3
"""Launches ONE single-op entry point of the C ABI a few times (bf16): the target of the rocprofv3 --pmc passes of
profiles/collect_counters.sh for the stride-2 / conv_trans kernels.
    op_kernel.py <fwd|dgrad|wgrad|convt_fwd|convt_dgrad|convt_wgrad> <cin> <cout> <input size> [stride = 2] [iters = 5]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import unet_studio_amd as U  # noqa: E402

E = U.engine
kind, cin, cout, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
stride = int(sys.argv[5]) if len(sys.argv) > 5 else 2
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream
bf = torch.bfloat16
nb = C.c_size_t()
E.check(E.lib.unet_op_scratch_bytes(cin, cout, n, n, n, C.byref(nb)))
sc = torch.empty(nb.value, dtype=torch.uint8, device=dev)
if kind.startswith("convt"):
    no = 2 * n
    x = torch.randn((n, n, n, cin), device=dev).to(bf)
    y = torch.randn((no, no, no, cout), device=dev).to(bf)
    w = torch.randn((cin, cout, 2, 2, 2), device=dev) * 0.1
    b = torch.zeros(cout, device=dev)
    dw, db = torch.zeros_like(w), torch.zeros_like(b)

    def run():
        if kind == "convt_fwd":
            E.check(E.lib.unet_op_convt_fwd(U.DTYPE_BF16, 0, x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), cin, cout, n, n, n, sc.data_ptr(), st))
        elif kind == "convt_dgrad":
            E.check(E.lib.unet_op_convt_bwd_data(U.DTYPE_BF16, 0, y.data_ptr(), w.data_ptr(), x.data_ptr(), cin, cout, n, n, n, sc.data_ptr(), st))
        else:
            E.check(E.lib.unet_op_convt_bwd_weight(U.DTYPE_BF16, 0, x.data_ptr(), y.data_ptr(), dw.data_ptr(), db.data_ptr(), cin, cout, n, n, n, sc.data_ptr(), st))
else:
    no = (n - 1) // stride + 1
    x = torch.randn((n, n, n, cin), device=dev).to(bf)
    y = torch.randn((no, no, no, cout), device=dev).to(bf)
    w = torch.randn((cout, cin, 3, 3, 3), device=dev) * 0.1
    b = torch.zeros(cout, device=dev)
    dw, db = torch.zeros_like(w), torch.zeros_like(b)

    def run():
        if kind == "fwd":
            E.check(E.lib.unet_op_conv3d_fwd(U.DTYPE_BF16, 0, x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), cin, cout, n, n, n, 3, stride, sc.data_ptr(), st))
        elif kind == "dgrad":
            E.check(E.lib.unet_op_conv3d_bwd_data(U.DTYPE_BF16, 0, y.data_ptr(), w.data_ptr(), x.data_ptr(), cin, cout, n, n, n, 3, stride, sc.data_ptr(), st))
        else:
            E.check(E.lib.unet_op_conv3d_bwd_weight(U.DTYPE_BF16, 0, x.data_ptr(), y.data_ptr(), dw.data_ptr(), db.data_ptr(), cin, cout, n, n, n, 3, stride,
                                                    sc.data_ptr(), st))
for _ in range(iters):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    run()
e1.record()
torch.cuda.synchronize()
print("%s %d->%d @%d^3: %.1f us per call (incl. the filter pack / slab reduce the entry point adds)" % (kind, cin, cout, n, e0.elapsed_time(e1) / iters * 1e3))
