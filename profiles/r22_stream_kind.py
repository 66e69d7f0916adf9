import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
U = importlib.import_module("unet-studio_amd")
m = U.UNet3d(1, 6, U.default_feature(6), device="cuda:0", dtype="bf16", seed=0)
tr = U.Trainer(m, U.TrainingParam(batch_size=1, epoch=100000, learning_rate=0.001), U.SyntheticVolumes(1, 6, (128, 128, 128), "cuda:0", cache=2), 0, 1)
def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): tr.step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for _ in range(10): tr.step()
print("default stream   %.3f %.3f" % (run(40), run(40)))
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for _ in range(10): tr.step()
    print("created stream   %.3f %.3f" % (run(40), run(40)))
hp = torch.cuda.Stream(priority=-1)
with torch.cuda.stream(hp):
    for _ in range(10): tr.step()
    print("high-priority    %.3f %.3f" % (run(40), run(40)))
