#!/usr/bin/env python3
"""How much longer the caller's stream's kernels take beside the side stream than alone.
   stretch.py <concurrent kernel_trace.csv> <solo kernel_trace.csv (UNET_NO_SIDE_STREAM=1)> [steps, default 8] [lines, default 40]
Kernels are matched by (name, grid, occurrence inside the step); durations are medians over the last steps of each trace.  For every
kernel of the caller's queue: solo duration, duration beside the side stream, and the side-stream kernel that overlapped it longest."""
import collections
import csv
import statistics
import sys


def steps_of(path, nsteps):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    sgd = [i for i, r in enumerate(rows) if "k_sgd" in r["Kernel_Name"]]
    out = []
    for k in range(len(sgd) - nsteps, len(sgd)):
        out.append(rows[sgd[k - 1] + 1: sgd[k] + 1])
    return out


def keyed(step):
    seen = collections.Counter()
    res = []
    for r in step:
        name = r["Kernel_Name"].split("(")[0]
        k0 = (name, r["Grid_Size_X"], r.get("Grid_Size_Y", ""))
        res.append((k0 + (seen[k0],), r))
        seen[k0] += 1
    return res


nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
nlines = int(sys.argv[4]) if len(sys.argv) > 4 else 40
conc, solo = steps_of(sys.argv[1], nsteps), steps_of(sys.argv[2], nsteps)
solo_d = collections.defaultdict(list)
for st in solo:
    for k, r in keyed(st):
        solo_d[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
conc_d = collections.defaultdict(list)
over = collections.defaultdict(collections.Counter)
qkey = "Queue_Id"
for st in conc:
    mainq = st[-1][qkey]                     # k_sgd runs on the caller's stream
    side = [r for r in st if r[qkey] != mainq]
    for k, r in keyed(st):
        if r[qkey] != mainq:
            continue
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        conc_d[k].append(e - s)
        for o in side:
            ov = min(e, int(o["End_Timestamp"])) - max(s, int(o["Start_Timestamp"]))
            if ov > 0:
                over[k][o["Kernel_Name"].split("(")[0][-40:] + " " + o["Grid_Size_X"]] += ov
tot_solo = tot_conc = 0.0
lines = []
for k, v in conc_d.items():
    if k not in solo_d:
        continue
    c, s = statistics.median(v) / 1e3, statistics.median(solo_d[k]) / 1e3
    tot_solo += s; tot_conc += c
    who = over[k].most_common(1)
    lines.append((c - s, s, c, k, who[0][0] if who else "-"))
lines.sort(reverse=True)
print("caller's-stream kernels matched: %d;  alone %.1f us, beside the side stream %.1f us, stretch %.1f us per step" % (len(lines), tot_solo, tot_conc, tot_conc - tot_solo))
byfam = collections.Counter()
for d, s, c, k, w in lines:
    byfam[k[0][-40:]] += d
print("stretch by kernel:", ", ".join("%s %.0f" % (n.split("::")[-1][:28], v) for n, v in byfam.most_common(12)))
for d, s, c, k, w in lines[:nlines]:
    print("  +%6.1f us  alone %6.1f  beside %6.1f  %-44s grid %-8s #%d   beside: %s" % (d, s, c, k[0][-44:], k[1], k[3], w))
