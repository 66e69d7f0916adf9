#!/usr/bin/env python3
"""How often does the engine's majority label differ from the numpy restatement's (libm last-bit differences on exact ties)?
Prints the mismatch counts of the parity cases of tests/test_gpu_augment.py."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_augment as T  # noqa: E402
from unet_studio_amd import augment as G  # noqa: E402
from oracle import augment_ref as R  # noqa: E402

tot = bad = 0
worst = 0.0
for case, (shape, ch, opt, is_label) in enumerate(T.CASES):
    for seed in (1, 2, 3):
        img, lab = T._sample(shape, ch, 4, seed)
        r = G.make_recipe(opt, shape, ch, is_label, seed * 7919 + case)
        gi, gl = T._run(r, img, lab)
        ri, rl = R.augment(r, img, lab)
        same = gl == rl if is_label else np.ones(rl.shape, bool)
        tot += same.size
        bad += int((~same).sum())
        worst = max(worst, float(np.abs(gi - ri)[:, same].max()))
print("label voxels %d, mismatching %d; worst image error %.3g" % (tot, bad, worst))
