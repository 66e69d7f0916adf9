#!/bin/bash
# HBM traffic of every kernel of the train step (two rocprofv3 --pmc passes over bench.py, --kernel-trace only), summed per kernel
# name: shows where the step's bytes go.  Run on the GPU box from the repo root: bash profiles/collect_step_traffic.sh
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp UNET_NO_SIDE_STREAM=1
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/step_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/step_$c -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-profile --no-kernels --no-cpp-host --batch 0 > $R/gpurun_out/step_$c.log 2>&1
done
python3 - "$R" <<'PY'
import csv, glob, json, sys, collections, re
R = sys.argv[1]
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.Counter(); n = collections.Counter()
    for f in glob.glob("%s/gpurun_out/step_%s/*/*counter_collection.csv" % (R, c)) + glob.glob("%s/gpurun_out/step_%s/*counter_collection.csv" % (R, c)):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c: continue
            k = re.sub(r"<.*", "", r["Kernel_Name"]).replace("void ", "").replace("unet::", "")
            acc[k] += float(r["Counter_Value"]); n[k] += 1
    tot[c] = (acc, n)
steps = 5.0   # 4 timed + 1 warm-up (no profile pass, no micro-benchmarks: --no-profile --no-kernels)
keys = set(tot["FETCH_SIZE"][0]) | set(tot["WRITE_SIZE"][0])
rows = []
for k in keys:
    fe = 2.0 * tot["FETCH_SIZE"][0][k] * 1024 / steps; wr = tot["WRITE_SIZE"][0][k] * 1024 / steps
    rows.append((fe + wr, fe, wr, tot["FETCH_SIZE"][1][k] / steps, k))
rows.sort(reverse=True)
print("HBM bytes per step (FETCH_SIZE doubled per MI355X_MICROARCH.md, + WRITE_SIZE), MB:  total %.0f" % (sum(r[0] for r in rows) / 1e6))
# what bench.py's roofline_step reports as counter_bytes (copy gpurun_out/step_hbm_traffic.json to profiles/)
json.dump({"hbm_bytes_per_step": sum(r[0] for r in rows), "read_bytes_per_step": sum(r[1] for r in rows), "write_bytes_per_step": sum(r[2] for r in rows),
           "note": "rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE in passes of their own over 5 whole train steps (128^3 bf16, side stream off), "
                   "profiles/collect_step_traffic.sh"}, open(R + "/gpurun_out/step_hbm_traffic.json", "w"), indent=1)
for t, fe, wr, n, k in rows[:40]:
    print("%9.1f MB  read %8.1f  write %8.1f  launches/step %5.1f  %s" % (t / 1e6, fe / 1e6, wr / 1e6, n, k[:70]))
PY
