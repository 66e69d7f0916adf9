# compares two dumps of profiles/dump_grads.py:  python3 profiles/cmp_grads.py a.npz b.npz
import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
print("loss", float(a["loss"]), float(b["loss"]))
for k in a.files:
    if k == "loss": continue
    x, y = a[k].astype(np.float64), b[k].astype(np.float64)
    den = max(np.abs(y).max(), 1e-30)
    print("%-40s max|d|/max %.3e   l2 %.4e vs %.4e" % (k, np.abs(x - y).max() / den, np.linalg.norm(x), np.linalg.norm(y)))
