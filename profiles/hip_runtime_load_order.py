"""Demonstrates why unet-studio_amd/engine.py imports torch before it loads libunet_hip.so (run on the GPU box):
  python3 profiles/hip_runtime_load_order.py torch_first    -> both calls OK, one libamdhip64 (torch's) in /proc/self/maps
  python3 profiles/hip_runtime_load_order.py engine_first   -> raw CDLL of the library before torch: /opt/rocm's libamdhip64 is
      loaded for it, torch then loads its own, and our runtime reports "no ROCm-capable device is detected".
Importing the package (not the raw CDLL) is always safe: engine.py fixes the order."""
import sys, os
sys.path.insert(0, os.getcwd())
if sys.argv[1] == "engine_first":
    import ctypes
    ctypes.CDLL(os.path.join(os.getcwd(), "unet-studio_amd", "libunet_hip.so"))
import torch
import unet_studio_amd as U
from unet_studio_amd import augment as G
x = torch.zeros(2*8*8*8, device="cuda:0"); l = torch.zeros(8*8*8, device="cuda:0")
r = G.make_recipe(None, (8,8,8), 2, True, 0)
try:
    G.augment(r, x, l); torch.cuda.synchronize(); print(sys.argv[1], "OK")
except Exception as e:
    print(sys.argv[1], "FAIL", e)
m = U.UNet3d(1, 3, "conv8,ks3,stride1+norm,leaky_relu\nconv16,ks3,stride2+norm,leaky_relu+conv_trans8,ks2,stride2\nconv8,ks3,stride1+norm,leaky_relu+conv3,ks1,stride1", device="cuda:0", dtype="fp32", seed=0)
try:
    o = m.forward(torch.rand(1, 1, 8, 8, 8, device="cuda:0")); torch.cuda.synchronize(); print("forward OK")
except Exception as e:
    print("forward FAIL", e)
print([l.strip().split()[-1] for l in open("/proc/self/maps") if "amdhip64" in l or "hsa-runtime" in l][::6])
