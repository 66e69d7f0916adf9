#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid) total/avg time, normalised per bench step."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
if steps <= 0:   # 0 = count them: one k_sgd launch per optimizer step (warm-up and bench.py's profile / batch passes included)
    steps = float(sum(1 for r in rows if "k_sgd" in r["Kernel_Name"])) or 1.0
agg = collections.defaultdict(lambda: [0, 0])
tot = 0
for r in rows:
    key = (r["Kernel_Name"].split("(")[0][-70:], r["Grid_Size_X"], r["Grid_Size_Y"])
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[key][0] += d
    agg[key][1] += 1
    tot += d
print("total kernel time %.3f ms (%.3f ms per optimizer step over %g steps = k_sgd launches in the trace when counted)" % (tot / 1e6, tot / 1e6 / steps, steps))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[: int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print("%8.3f ms/step  avg %8.3f ms  n=%4d  %s" % (v[0] / 1e6 / steps, v[0] / v[1] / 1e6, v[1], k))
