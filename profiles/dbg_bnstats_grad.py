#!/usr/bin/env python3
"""One forward + backward of a small bf16 network with the norm-backward statistics fused into the dgrad epilogue and separate (fresh
processes: the switch is read once): per-parameter difference of the gradients.  Only the summation order of the statistics differs."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path.insert(0, %r)
import numpy as np, torch
import unet_studio_amd as U
n = int(sys.argv[2])
arch = U.default_feature(6) if sys.argv[3] == "default" else ("conv16,ks3,stride1+norm,leaky_relu+conv16,ks3,stride1+norm,leaky_relu\n"
        "conv32,ks3,stride2+norm,leaky_relu+conv32,ks3,stride1+norm,leaky_relu+conv_trans16,ks2,stride2\n"
        "conv16,ks3,stride1+norm,leaky_relu+conv16,ks3,stride1+norm,leaky_relu+conv6,ks1,stride1")
m = U.UNet3d(1, 6, arch, device="cuda:0", dtype="bf16", seed=0)
src = U.SyntheticVolumes(1, 6, (n, n, n), "cuda:0", cache=2)
x, t = src(0)
m.zero_grad() if hasattr(m, "zero_grad") else None
m.forward_backward(x, t)
torch.cuda.synchronize()
np.save(sys.argv[1], m.flat_grads.cpu().numpy())
import json
json.dump([[nm, int(np.prod(s))] for nm, s in zip(m.plan_for((n, n, n)).param_names, m.plan_for((n, n, n)).param_shapes)], open(sys.argv[1] + ".json", "w"))
''' % ROOT
n = sys.argv[1] if len(sys.argv) > 1 else "16"
arch = sys.argv[2] if len(sys.argv) > 2 else "small"
g = {}
for fused in (1, 0):
    env = dict(os.environ)
    if not fused:
        env["UNET_NO_DGRAD_BNSTATS"] = "1"
    path = "/tmp/g_f%d.npy" % fused
    subprocess.check_call([sys.executable, "-c", CHILD, path, n, arch], env=env)
    g[fused] = np.load(path)
import json
names = json.load(open("/tmp/g_f1.npy.json"))
off = 0
for nm, cnt in names:
    a, b = g[1][off:off + cnt], g[0][off:off + cnt]
    off += cnt
    d = np.abs(a - b).max()
    if d > 0:
        print("%-40s n %8d  max|diff| %.3e  max|ref| %.3e  rel %.2e" % (nm, cnt, d, np.abs(b).max(), d / (np.abs(b).max() + 1e-30)))
print("total max |diff| %.3e" % np.abs(g[1] - g[0]).max())
