# dumps logits / loss / every parameter gradient of one 128^3 (or N^3) bf16 training step to an .npz:  python3 profiles/dump_grads.py out.npz [n]
# (run twice with different switches -- e.g. UNET_NO_SLIDING_WINDOW=1 -- and compare with profiles/cmp_grads.py)
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
U = importlib.import_module("unet-studio_amd.unet3d")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import aten_ref as A
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
torch.manual_seed(0)
ref = A.UNet3dRef(1, 6, A.default_feature(6))
m = U.UNet3d(1, 6, A.default_feature(6), device="cuda:0", dtype="bf16")
m.load_parameters([p.detach().numpy() for p in ref.parameters()])
x, t = A.synthetic_sample(1, 6, (n, n, n), 1)
x, t = x.cuda(), t.cuda()
plan = m.plan_for(x.shape[2:]); ws = m._workspace(plan)
outs = m._run_forward(plan, ws, x, 1)
losses, gouts = m.loss(outs, t)
m._run_backward(plan, ws, gouts)
torch.cuda.synchronize()
d = {"loss": float(losses[0])}
for k, o in enumerate(outs): d["logits%d" % k] = o.float().cpu().numpy()
names = [nm for nm, _ in ref.named_parameters()]
for nm, g in zip(names, m.grads()): d["g:" + nm] = g.float().cpu().numpy()
np.savez(sys.argv[1], **d)
print("loss", d["loss"])
