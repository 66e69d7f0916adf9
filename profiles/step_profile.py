#!/usr/bin/env python3
"""Per-op kernel time of the train step from the engine's own HIP-event profiler (unet_profile_begin/end: every op bracketed on
ONE stream, side stream off): where the step's time goes by layer and pass.   step_profile.py [size] [in_channels] [steps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import unet_studio_amd as U  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cin = int(sys.argv[2]) if len(sys.argv) > 2 else 1
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = "cuda:0"
m = U.UNet3d(cin, 6, U.default_feature(6), device=dev, dtype="bf16", seed=0)
src = U.SyntheticVolumes(cin, 6, (n, n, n), dev, cache=2)
tr = U.Trainer(m, U.TrainingParam(batch_size=1, epoch=10000, learning_rate=0.001), lambda i: src(i % 2))
for _ in range(3):
    tr.step()
torch.cuda.synchronize()
with U.engine.profile() as pr:
    for _ in range(steps):
        tr.step()
    torch.cuda.synchronize()
plan = m.plan_for((n, n, n))
ops = plan.ops()
per, fam, cnt = {}, {}, 0
for op, cat, ms in pr.records:
    per[(op, cat)] = per.get((op, cat), 0.0) + ms / steps
    fam[cat] = fam.get(cat, 0.0) + ms / steps
    cnt += 1
print("brackets per step: %d;  by family (ms/step): %s;  total %.3f" % (cnt // steps, {k: round(v, 3) for k, v in sorted(fam.items())}, sum(fam.values())))
rows = []
for (op, cat), ms in per.items():
    o = ops[op] if op >= 0 else None
    name = o["name"] if o else "(batched)"
    fl = 0.0
    if o and o["kind"] == 1:
        v = o["out_dims"]
        fl = 2.0 * o["ks"] ** 3 * o["cin"] * o["cout"] * v[0] * v[1] * v[2]
    elif o and o["kind"] == 2:
        v = o["in_dims"]
        fl = 2.0 * 8 * o["cin"] * o["cout"] * v[0] * v[1] * v[2]
    frac = fl / (ms * 1e-3) / 2.5e15 if (ms > 0 and cat in ("conv_fwd", "dgrad", "wgrad")) else 0.0
    rows.append((ms, cat, name, frac, o))
rows.sort(key=lambda r: -r[0])
for ms, cat, name, frac, o in rows[: int(os.environ.get("TOP", "70"))]:
    dims = ("%dx%dx%d" % tuple(o["out_dims"])) if o and o["out_dims"][0] else (("%dx%dx%d" % tuple(o["in_dims"])) if o else "")
    print("%8.4f ms  %-9s %-44s %-12s %s" % (ms, cat, name, dims, ("mfma %.3f" % frac) if frac else ""))
