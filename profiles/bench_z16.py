#!/usr/bin/env python3
"""16-channel-chunk stride-1 layers at full resolution through the C ABI (op level: filter pack + kernel): forward and dgrad of 16->16,
dgrad of 32->16 (16 -> 32 rows).  Run twice: default and UNET_NO_CONV_Z16=1 (halo-tile kernel)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_kernels import conv_case  # noqa: E402

if __name__ == "__main__":
    print("variant env:", {k: v for k, v in os.environ.items() if k.startswith("UNET_")})
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    conv_case(16, 16, n, fused=False, what="fwd")
    conv_case(16, 16, n, what="dgrad")
    conv_case(32, 16, n, what="dgrad")
    conv_case(16, 32, n, fused=False, what="fwd")
