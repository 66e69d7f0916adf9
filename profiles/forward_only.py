#!/usr/bin/env python3
"""Forward-only timing (BASELINE.json configs[1]: full 5-level UNet3d forward on a 128^3 single-channel volume, fp32; and the
same in bf16), eval mode as evaluate.cpp runs it, HIP events on the launch stream.  One JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_studio_amd as U  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
out = {}
for dt in ("fp32", "bf16"):
    m = U.UNet3d(1, 6, U.default_feature(6), device="cuda:0", dtype=dt, seed=0)
    m.prepare_for_inference()
    x = torch.rand(1, 1, n, n, n, device="cuda:0")
    with torch.no_grad():
        for _ in range(3):
            m.forward(x)
        torch.cuda.synchronize()
        # an inference run: the weights are frozen, volumes 2.. reuse the filter packs of the first (evaluate.py does the same)
        pc = os.environ.get("UNET_FWD_REPACK") is None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 10
        e0.record()
        for _ in range(it):
            m.forward(x, packs_current=pc)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    fl = m.plan_for((n, n, n)).flops_fwd
    out[dt] = {"ms_per_forward": ms, "voxels_per_s": n ** 3 / (ms * 1e-3), "TFLOP_per_s": fl / (ms * 1e-3) / 1e12}
print(json.dumps({"workload": "default arch forward, in=1 out=6, %d^3, eval mode" % n, **out}))
