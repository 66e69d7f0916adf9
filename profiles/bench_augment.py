#!/usr/bin/env python3
"""Times the on-GPU augmentation (include/unet_augment.h) at BASELINE.json configs[4]'s sample size -- 2 channels of 256^3 --
with HIP events on the launch stream, and prints one JSON line: ms per sample, voxels/s, and the HBM rate against the
passes' algorithmic bytes (every pass reads / writes each volume it touches once; gathers counted once per source volume).
  python profiles/bench_augment.py [--size 256] [--channels 2] [--iters 20]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_studio_amd as U  # noqa: E402
from unet_studio_amd import augment as G  # noqa: E402


def options(level, **over):
    o = dict(G.DEFAULT_OPTIONS)
    for k in ("cropping", "truncation_z", "downsample_x", "downsample_y", "downsample_z", "noise", "ambient", "diffuse", "specular",
              "distortion", "zero_background", "rubber_stamping", "perlin_texture"):
        o[k] = level
    o.update(over)
    return o


def volumes_moved(r):
    """fp32 volumes (of D*H*W voxels) read + written per sample by the passes this recipe runs."""
    c = r["channels"]
    v = 0.0
    if r["downsample"]:
        low = r["low_dims"][0] * r["low_dims"][1] * r["low_dims"][2] / float(r["dims"][0] * r["dims"][1] * r["dims"][2])
        v += c * 2 * (1 + low)
    if any(r[k] for k in ("crop", "trunc_top", "trunc_bottom", "noise", "ambient", "diffuse", "specular")):
        v += 2 * (c + 1)
    v += 2 * (c + 1)                      # view: gather image + label, write out + out_label
    if r["is_label"] and not r["zero_background"] and (r["rubber"] or r["perlin"]):
        v += (c + 1 if r["rubber"] else 0)            # maxima: stamps gather the pre-view image + label
        v += c + 1 + (c + 1 if r["rubber"] else 0) + c   # blend: out, out_label, stamp gathers, write out
    v += 2 * (c + 1)                      # final
    return v


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--channels", type=int, default=2)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dev = "cuda:0"
    n, ch = a.size, a.channels
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    x0 = torch.rand((ch, n, n, n), device=dev, generator=g)
    zz = torch.arange(n, device=dev, dtype=torch.float32)
    rr = ((zz[:, None, None] - n / 2) / (n * 0.35)) ** 2 + ((zz[None, :, None] - n / 2) / (n * 0.3)) ** 2 + \
         ((zz[None, None, :] - n / 2) / (n * 0.4)) ** 2
    l0 = (rr < 1).float() + (rr < 0.5).float() + (rr < 0.2).float()
    out = {"sample": "%d channels of %d^3 fp32 + label, resident in HBM" % (ch, n), "cases": {}}
    for name, opt in (("all_stages_blend", options(4, zero_background=0)), ("all_stages_zero_background", options(4)),
                      ("view_only", options(0)), ("shipped_options_mean_of_seeds", None)):
        recipes = [G.make_recipe(opt, (n, n, n), ch, True, s) for s in (range(8) if opt is None else [1])]
        scratch = torch.empty(max(G.scratch_bytes(r) for r in recipes), dtype=torch.uint8, device=dev)
        structs = [G.to_struct(r) for r in recipes]
        x, l = x0.clone(), l0.clone()
        for s in structs:
            G.augment(s, x.view(-1), l.view(-1), scratch)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tot, cnt = 0.0, 0
        for _ in range(a.iters):
            x.copy_(x0)
            l.copy_(l0)
            for s in structs:
                e0.record()
                G.augment(s, x.view(-1), l.view(-1), scratch)
                e1.record()
                torch.cuda.synchronize()
                tot += e0.elapsed_time(e1)
                cnt += 1
        ms = tot / cnt
        vols = sum(volumes_moved(r) for r in recipes) / len(recipes)
        gb = vols * n ** 3 * 4 / 1e9
        out["cases"][name] = {"ms_per_sample": ms, "voxels_per_s": n ** 3 / (ms * 1e-3), "algorithmic_GB": gb,
                              "GB_per_s": gb / (ms * 1e-3), "frac_of_8TBps": gb / (ms * 1e-3) / 8000.0}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
