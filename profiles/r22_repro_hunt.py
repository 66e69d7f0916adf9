# bit-reproducibility of the step with the ticket kernels (kernels_mfma_deep.hip): N fresh processes, default architecture at 32^3 and 128^3,
# SHA-1 of the flat gradient after one forward + backward; every process must print the same digest per size
import hashlib, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
U = importlib.import_module("unet-studio_amd")
out = []
for n in (32, 128):
    m = U.UNet3d(1, 6, U.default_feature(6), device="cuda:0", dtype="bf16", seed=0)
    x, t = U.SyntheticVolumes(1, 6, (n, n, n), "cuda:0", cache=2)(0)
    for _ in range(3):       # three steps on one workspace: the counters must be back at zero every time
        m.zero_grad() if hasattr(m, "zero_grad") else None
        m.forward_backward(x, t)
    torch.cuda.synchronize()
    out.append("%d:%s" % (n, hashlib.sha1(m.flat_grads.cpu().numpy().tobytes()).hexdigest()[:12]))
print(" ".join(out))
