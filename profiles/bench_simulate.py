#!/usr/bin/env python3
"""Times simulate_modality (include/unet_augment.h) on a resident 256^3 volume with HIP events.  One JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_studio_amd as U  # noqa: E402,F401
from unet_studio_amd import augment as G  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = torch.Generator(device="cuda:0")
g.manual_seed(0)
x0 = torch.rand((n, n, n), device="cuda:0", generator=g)
lab = (torch.rand((n, n, n), device="cuda:0", generator=g) * 6).floor()
out = {"sample": "%d^3 fp32 volume + label volume, resident in HBM" % n}
for name, ml in (("with_label", 5), ("without_label", None)):
    r = G.sim_to_struct(G.make_simulate_recipe((n, n, n), ml, 3))
    x = x0.clone()
    sc = G.simulate(r, x, lab if ml is not None else None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for _ in range(20):
        x.copy_(x0)
        e0.record()
        G.simulate(r, x, lab if ml is not None else None, sc)
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    ms = tot / 20
    vols = 2 + 2 + (3 if ml is not None else 2) + 2      # smooth (r+w) x2, remap (t1w r/w, tissue, label), final (r+w)
    gb = vols * n ** 3 * 4 / 1e9
    out[name] = {"ms": ms, "voxels_per_s": n ** 3 / (ms * 1e-3), "algorithmic_GB": gb, "GB_per_s": gb / (ms * 1e-3)}
print(json.dumps(out))
