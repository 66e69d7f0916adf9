import os, subprocess, sys, numpy as np
sys.path.insert(0, "profiles")
src = open("profiles/dbg_bnstats_grad.py").read()
child = src[src.index("CHILD = r'''")+len("CHILD = r'''"):src.index("''' % ROOT")] % os.getcwd()
outs = []
for k in range(3):
    p = "/tmp/det%d.npy" % k
    subprocess.check_call([sys.executable, "-c", child, p, "16", "small"])
    outs.append(np.load(p))
print("run-to-run max diff:", np.abs(outs[0]-outs[1]).max(), np.abs(outs[0]-outs[2]).max())
