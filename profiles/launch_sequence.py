#!/usr/bin/env python3
"""The launch sequence of ONE optimizer step out of a rocprofv3 --kernel-trace CSV (best taken with UNET_NO_SIDE_STREAM=1, so that the
order is the engine's issue order): kernel, blocks (x, y), threads per block, LDS bytes, VGPRs (+ AGPRs), duration.  This is the
"what runs by default" listing DESIGN.md section 4 summarises per level.     launch_sequence.py <kernel_trace.csv> [step index from the end = 2]"""
import csv
import re
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kind"] == "KERNEL_DISPATCH"]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
sgd = [i for i, r in enumerate(rows) if "k_sgd" in r["Kernel_Name"]]
if len(sgd) < back + 1:
    sys.exit("need at least %d optimizer steps in the trace" % (back + 1))
lo, hi = sgd[-back - 1] + 1, sgd[-back] + 1
t0 = int(rows[lo]["Start_Timestamp"])
tot = 0
print("# launches %d .. %d of the trace: one optimizer step (forward, loss, backward, update); t = start offset" % (lo, hi - 1))
print("%4s %9s %8s  %-58s %12s %5s %7s %9s" % ("#", "t us", "dur us", "kernel", "blocks", "thr", "LDS B", "VGPR+AGPR"))
for k, r in enumerate(rows[lo:hi]):
    name = re.sub(r"^void unet::", "", r["Kernel_Name"].split("(")[0])
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    bx = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
    by = int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"]))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    print("%4d %9.1f %8.1f  %-58s %12s %5d %7s %9s" % (k, (int(r["Start_Timestamp"]) - t0) / 1e3, d, name[:58], "%d x %d" % (bx, by), wg, r["LDS_Block_Size"],
                                                      "%s+%s" % (r["VGPR_Count"], r["Accum_VGPR_Count"])))
print("# %d launches, %.1f us of kernel time, %.1f us from the first start to the last end" %
      (hi - lo, tot, (int(rows[hi - 1]["End_Timestamp"]) - t0) / 1e3))
