#!/usr/bin/env python3
"""One train step out of a rocprofv3 --kernel-trace CSV as a time line: per kernel its start offset inside the step, its duration,
the hardware queue it ran on and the gap to the previous kernel of that queue; then per queue busy time and the wall time of the step.
   timeline.py <kernel_trace.csv> [step index from the end, default 2] [max lines]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
which = int(sys.argv[2]) if len(sys.argv) > 2 else 2
maxl = int(sys.argv[3]) if len(sys.argv) > 3 else 400
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sgd = [i for i, r in enumerate(rows) if "k_sgd" in r["Kernel_Name"]]
if len(sgd) < which + 1:
    sys.exit("not enough steps in the trace")
lo, hi = sgd[-which - 1] + 1, sgd[-which] + 1
step = rows[lo:hi]
t0 = int(rows[lo - 1]["End_Timestamp"])
qkey = "Queue_Id" if "Queue_Id" in step[0] else ("Stream_Id" if "Stream_Id" in step[0] else None)
last_end = {}
busy = collections.defaultdict(int)
print("step of %d kernels, wall %.1f us (end of previous k_sgd -> end of this k_sgd)" % (len(step), (int(step[-1]["End_Timestamp"]) - t0) / 1e3))
for n, r in enumerate(step):
    q = r[qkey] if qkey else "0"
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - last_end[q]) / 1e3 if q in last_end else (s - t0) / 1e3
    last_end[q] = e
    busy[q] += e - s
    if n < maxl:
        name = r["Kernel_Name"].split("(")[0][-58:]
        print("%9.1f us  +%7.1f  gap %6.1f  q%-3s %-58s grid %s x %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, q, name, r["Grid_Size_X"], r.get("Grid_Size_Y", "")))
for q, b in sorted(busy.items()):
    print("queue %s busy %.1f us" % (q, b / 1e3))
# time covered by at least one kernel / by two or more
ev = []
for r in step:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
cov1 = cov2 = 0
depth, prev = 0, ev[0][0]
for t, d in ev:
    if depth >= 1: cov1 += t - prev
    if depth >= 2: cov2 += t - prev
    depth += d; prev = t
print("covered by >= 1 kernel %.1f us, by >= 2 kernels %.1f us" % (cov1 / 1e3, cov2 / 1e3))
