"""GPU (-m gpu): the C++ drop-in (include/unet.hpp + unet_host.cpp over libtorch, the reference's own language) run as a
compiled binary: parameter names/order, Module::to, forward parity against a torch::nn network assembled as unet.cpp
does, autograd backward, the reference's optimizer flow (zero_grad -> None grads), prepare_for_inference, copy_from /
add_gradient_from."""
import os
import subprocess

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "unet-studio_amd")


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_cpp_drop_in(mode):
    exe = os.path.join(PKG, "test_unet_hpp")
    if not os.path.exists(exe):
        subprocess.check_call(["bash", os.path.join(PKG, "csrc", "build_host.sh")])
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = PKG + ":" + os.path.join(os.path.dirname(torch.__file__), "lib") + ":" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([exe, mode], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "OK " + mode in r.stdout
