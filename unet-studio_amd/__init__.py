"""MI355X-native engine for the UNet3d hot path of UNet-Studio (unet.hpp / unet.cpp / train.cpp step loop).

Host-side mirror of the reference's operator surface over libunet_hip.so (include/unet_hip.h).
Importing this package loads the HIP library; if it is missing the import fails -- there is no fallback.
"""
from . import engine  # noqa: F401  (raises ImportError when libunet_hip.so is absent)
from .engine import DTYPE_BF16, DTYPE_F32, IMPL_AUTO, IMPL_DIRECT, Plan, UNetError  # noqa: F401
from .unet3d import SGD, UNet3d  # noqa: F401
from .train import SyntheticVolumes, Trainer, TrainingParam  # noqa: F401
from . import augment  # noqa: F401
from .evaluate import EvaluateUNet  # noqa: F401
from .augment import AugmentedVolumes, PrefetchedVolumes, visual_perception_augmentation  # noqa: F401
from . import nz  # noqa: F401
from .nz import NzError  # noqa: F401


def save_to_file(model, file_name):
    """bool save_to_file(UNet3d& model, const char* file_name) -- main.cpp:207-233 (`.nz`: gzip + MATLAB Level-4 records, nz.py)"""
    return nz.save_to_file(model, file_name)


def load_from_file(file_name, device="cuda:0", dtype="bf16"):
    """load_from_file (main.cpp:157-206): builds UNet3d(channels[0], channels[1], architecture) on `device` and fills
    parameters() from tensor0..N; raises NzError with the reference's messages."""
    return nz.load_from_file(file_name, lambda i, o, a: UNet3d(i, o, a, device=device, dtype=dtype))


def default_feature(out_count):
    """The reference's default architecture string (train.cpp:1054-1069)."""
    out = "conv%d,ks1,stride1" % out_count
    nl = "norm,leaky_relu"
    enc = []
    for i, c in enumerate([16, 32, 64, 128, 256, 256]):
        enc.append("conv%d,ks3,stride%d+%s+conv%d,ks3,stride1+%s" % (c, 1 if i == 0 else 2, nl, c, nl))
    enc[-1] += "+conv_trans256,ks2,stride2"
    dec = ["conv%d,ks3,stride1+%s+conv%d,ks3,stride1+%s+%s+conv_trans%d,ks2,stride2" % (c, nl, c, nl, out, up)
           for c, up in [(256, 128), (128, 64), (64, 32), (32, 16)]]
    dec.append("conv16,ks3,stride1+%s+conv16,ks3,stride1+%s+%s" % (nl, nl, out))
    return "\n".join(enc + dec)
