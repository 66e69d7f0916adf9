# one process = 5 warm-up steps + 40 free-running timed steps (events around each, no host sync inside): prints the slowest steps
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
U = importlib.import_module("unet-studio_amd")
m = U.UNet3d(1, 6, U.default_feature(6), device="cuda:0", dtype="bf16", seed=0)
tr = U.Trainer(m, U.TrainingParam(batch_size=1, epoch=100000, learning_rate=0.001), U.SyntheticVolumes(1, 6, (128, 128, 128), "cuda:0", cache=2), 0, 1)
for _ in range(5): tr.step()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
host = []
t0 = time.perf_counter()
ev[0].record()
for i in range(40):
    h0 = time.perf_counter(); tr.step(); host.append((time.perf_counter() - h0) * 1e3)
    ev[i + 1].record()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 40 * 1e3
gpu = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(40)])
host = np.array(host)
print("wall %.3f ms/step; gpu interval median %.3f max %.3f at step %d; host enqueue median %.3f max %.3f at step %d" %
      (wall, np.median(gpu), gpu.max(), int(gpu.argmax()), np.median(host), host.max(), int(host.argmax())))
