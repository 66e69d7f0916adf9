#!/usr/bin/env python3
"""Launches only the dominant kernel of the step (decode0.0: conv3d 32->16 3x3x3 @128^3, bf16, with the norm-statistics
epilogue) a few times: the target of the rocprofv3 --pmc passes of profiles/collect_traffic.sh."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import unet_studio_amd as U  # noqa: E402

dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
flops, sec = bench.dominant_kernel(U, 128, dt, iters=5)
print("dominant kernel: %.3f ms per launch, %.1f TFLOP/s" % (sec * 1e3, flops / sec / 1e12))
