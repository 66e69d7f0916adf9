#!/usr/bin/env python3
"""The inference loop (evaluate.cpp:211-246 through unet-studio_amd/evaluate.py) end to end from host buffers: H2D of the volume,
forward, D2H of the 6-channel logits -- the PCIe-inclusive rate of BASELINE.json configs[1]'s workload.  One JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_studio_amd as U  # noqa: E402

n, nvol = 128, 16
rs = np.random.RandomState(0)
ios = [[rs.rand(n, n, n).astype(np.float32)] for _ in range(nvol)]
out = {"workload": "%d volumes of %d^3, in=1 out=6, default arch, host buffer -> logits in a host buffer" % (nvol, n)}
for dt in ("bf16", "fp32"):
    m = U.UNet3d(1, 6, U.default_feature(6), device="cuda:0", dtype=dt, seed=0)
    ev = U.EvaluateUNet(m)
    ev.start(ios[:2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = ev.start(ios)
    dtm = (time.perf_counter() - t0) / nvol
    assert not ev.aborted and res[0][0].shape == (6 * n, n, n)
    out[dt] = {"ms_per_volume": dtm * 1e3, "voxels_per_s": n ** 3 / dtm}
print(json.dumps(out))
