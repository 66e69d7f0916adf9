"""How far apart are the parameters of a 2-rank and a 1-rank fp32 run of tests/test_gpu_dist.py's scenario (summation order of the
micro-step gradients is the only difference)?"""
import os, sys
import numpy as np, torch
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_dist as T

if __name__ == "__main__":
    mgr = mp.Manager()
    single, out = mgr.dict(), mgr.dict()
    mp.spawn(T._run, args=(1, 0, 3, 4, "fp32", single), nprocs=1, join=True)
    mp.spawn(T._run, args=(2, T._free_port(), 3, 4, "fp32", out), nprocs=2, join=True)
    a, b = single[0][0], out[0][0]
    d = np.abs(a - b)
    i = int(d.argmax())
    print("max |diff| %.3g at %d (values %.6g %.6g); ranks equal: %s; > 1e-5: %d of %d" % (d.max(), i, a[i], b[i], np.array_equal(out[0][0], out[1][0]), int((d > 1e-5).sum()), d.size))
