"""Where does the fp32 engine's gradient differ from fp64 ATen on a non-cubic case?  Architecture variants + dL/dx."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import unet_studio_amd as U
from oracle import aten_ref as A
import test_gpu_parity as T
V0 = T.ARCH_NONCUBIC
V1 = V0.replace("+norm,leaky_relu", ",leaky_relu")
V2 = V0.replace("norm,leaky_relu", "norm")
V3 = V0.replace("norm,leaky_relu", "bnorm,leaky_relu")
V4 = V0.replace("leaky_relu", "elu")
import itertools
for size, (name, arch) in itertools.product([(16, 16, 36), (24, 40, 56), (20, 36, 12), (8, 16, 132)], (("V0 norm+leaky", V0), ("V4 norm+elu", V4))):
    torch.manual_seed(3)
    ref = A.UNet3dRef(2, 5, arch); ref.train()
    x, t = A.synthetic_sample(2, 5, size, 11)
    params = [p.detach().numpy().copy() for p in ref.parameters()]
    names = [n for n, _ in ref.named_parameters()]
    ref = ref.double()
    xr = x.double().requires_grad_(True)
    outs_ref = ref(xr)
    loss_ref, _ = A.deep_supervision_loss(outs_ref, t, 5); loss_ref.backward()
    gref = [p.grad.numpy() for p in ref.parameters()]
    gmax = max(np.abs(g).max() for g in gref)
    m = U.UNet3d(2, 5, arch, device="cuda:0", dtype="fp32", impl=U.IMPL_AUTO)
    m.load_parameters(params)
    xd, td = x.to("cuda:0"), t.to("cuda:0")
    plan = m.plan_for(xd.shape[2:]); ws = m._workspace(plan)
    outs = m._run_forward(plan, ws, xd, 1)
    losses, gouts = m.loss(outs, td)
    gx = torch.zeros_like(xd)
    m._run_backward(plan, ws, gouts, gx)
    errs = sorted([(float(np.abs(g.cpu().numpy() - r).max() / gmax), n) for g, r, n in zip(m.grads(), gref, names)], reverse=True)
    ex = float((gx.cpu().double() - xr.grad).abs().max() / xr.grad.abs().max())
    print(size, name, "dL/dx err %.2g;" % ex, ["%.2g %s" % e for e in errs[:4]])
