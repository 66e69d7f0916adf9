#!/usr/bin/env python3
"""Weight-gradient launches of the default architecture's 3x3x3 layers through unet_op_conv3d_bwd_weight (kernel + slab reduce):
time per launch and fraction of the bf16 MFMA peak.  UNET_NO_WGRAD_Z=1 selects the halo-tile kernel for an A/B in a second run."""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import unet_studio_amd as U  # noqa: E402

E = U.engine
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream
SHAPES = [(16, 16, 128, 1), (32, 16, 128, 1), (32, 32, 64, 1), (64, 32, 64, 1), (64, 64, 32, 1), (128, 64, 32, 1), (128, 128, 16, 1), (256, 128, 16, 1),
          (256, 256, 8, 1), (512, 256, 8, 1), (256, 256, 4, 1), (16, 32, 128, 2), (32, 64, 64, 2), (64, 128, 32, 2), (128, 256, 16, 2), (256, 256, 8, 2)]
# conv_trans (cin, cout, coarse n): decode_tail1..4 and encode5.6 of the default architecture
CONVT = [(32, 16, 64), (64, 32, 32), (128, 64, 16), (256, 128, 8), (256, 256, 4)]
out = []
for cin, cout, n, stride in SHAPES:
    no = n // stride
    x = torch.randn((n, n, n, cin), device=dev).to(torch.bfloat16)
    dy = torch.randn((no, no, no, cout), device=dev).to(torch.bfloat16)
    dw = torch.zeros((cout, cin, 3, 3, 3), device=dev)
    db = torch.zeros(cout, device=dev)
    nb = C.c_size_t()
    E.check(E.lib.unet_op_scratch_bytes(cin, cout, n, n, n, C.byref(nb)))
    sc = torch.empty(nb.value, dtype=torch.uint8, device=dev)

    def run():
        E.check(E.lib.unet_op_conv3d_bwd_weight(U.DTYPE_BF16, U.IMPL_AUTO, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), cin, cout,
                                                n, n, n, 3, stride, sc.data_ptr(), st))
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 20
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * 27 * cin * cout * no ** 3
    byts = (n ** 3 * cin + no ** 3 * cout) * 2
    rec = {"shape": "%d->%d @%d^3 s%d" % (cin, cout, n, stride), "ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 1), "mfma_frac": round(fl / ms / 1e9 / 2500, 3),
           "algorithmic_GBps": round(byts / ms / 1e6, 1)}
    out.append(rec)
    print(json.dumps(rec), flush=True)

for cin, cout, n in CONVT:
    x = torch.randn((n, n, n, cin), device=dev).to(torch.bfloat16)
    dy = torch.randn((2 * n, 2 * n, 2 * n, cout), device=dev).to(torch.bfloat16)
    dw = torch.zeros((cin, cout, 2, 2, 2), device=dev)
    db = torch.zeros(cout, device=dev)
    nb = C.c_size_t()
    E.check(E.lib.unet_op_scratch_bytes(cin, cout, n, n, n, C.byref(nb)))
    sc = torch.empty(nb.value, dtype=torch.uint8, device=dev)

    def run():
        E.check(E.lib.unet_op_convt_bwd_weight(U.DTYPE_BF16, U.IMPL_AUTO, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), cin, cout,
                                               n, n, n, sc.data_ptr(), st))
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 20
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    byts = (n ** 3 * cin + 8 * n ** 3 * cout) * 2
    rec = {"shape": "conv_trans %d->%d @%d^3 (wgrad + bias grad)" % (cin, cout, n), "ms": round(ms, 4), "algorithmic_GBps": round(byts / ms / 1e6, 1)}
    print(json.dumps(rec), flush=True)
