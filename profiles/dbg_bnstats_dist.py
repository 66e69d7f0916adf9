#!/usr/bin/env python3
"""How far the parameters of tests/test_gpu_dist.py's three-step bf16 run move between (a) the fused / separate norm-backward statistics
and (b) one rank / two ranks, each in fresh processes (the switches are read once per process)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch.multiprocessing as mp
import test_gpu_dist as T
world = int(sys.argv[1]); out_path = sys.argv[2]
mgr = mp.Manager(); out = mgr.dict()
mp.spawn(T._run, args=(world, T._free_port() if world > 1 else 0, 3, 4, "bf16", out), nprocs=world, join=True)
np.save(out_path, out[0][0])
''' % (ROOT, ROOT)

res = {}
for fused in (1, 0):
    for world in (1, 2):
        env = dict(os.environ)
        if not fused:
            env["UNET_NO_DGRAD_BNSTATS"] = "1"
        path = "/tmp/p_f%d_w%d.npy" % (fused, world)
        subprocess.check_call([sys.executable, "-c", CHILD, str(world), path], env=env)
        res[(fused, world)] = np.load(path)
for a, b in (((1, 1), (0, 1)), ((1, 2), (1, 1)), ((0, 2), (0, 1)), ((1, 2), (0, 2))):
    d = np.abs(res[a] - res[b])
    print("fused,world %s vs %s: max |diff| %.3e, > 1e-4: %d, > 5e-5: %d of %d" % (a, b, d.max(), int((d > 1e-4).sum()), int((d > 5e-5).sum()), d.size))
