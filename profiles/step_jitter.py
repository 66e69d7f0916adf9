# per-step wall time of the train step (each step synchronised): is a slow bench run many slow steps or one stall?
#   python3 profiles/step_jitter.py [steps]     (UNET_NO_DEEP_KERNELS=1 for the other kernel set)
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
U = importlib.import_module("unet-studio_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
m = U.UNet3d(1, 6, U.default_feature(6), device="cuda:0", dtype="bf16", seed=0)
tr = U.Trainer(m, U.TrainingParam(batch_size=1, epoch=100000, learning_rate=0.001), U.SyntheticVolumes(1, 6, (128, 128, 128), "cuda:0", cache=2), 0, 1)
for _ in range(10): tr.step()
torch.cuda.synchronize()
# (a) synchronised steps: GPU time of every single step
ts = []
for _ in range(n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); tr.step(); e1.record(); e1.synchronize()
    ts.append(e0.elapsed_time(e1))
ts = np.array(ts)
print("synchronised steps: median %.3f ms  p99 %.3f  max %.3f  steps > 1.5 x median: %d of %d" % (np.median(ts), np.percentile(ts, 99), ts.max(), int((ts > 1.5 * np.median(ts)).sum()), n))
# (b) free-running blocks of 30 steps, as bench.py times them
bl = []
for _ in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): tr.step()
    torch.cuda.synchronize(); bl.append((time.perf_counter() - t0) / 30 * 1e3)
print("free-running blocks of 30: " + " ".join("%.3f" % b for b in bl))
