#!/bin/bash
# HBM traffic of the fp32 engine's dominant kernel (k_conv_f32_mfma, conv3d 32->16 3x3x3 @128^3) from PMC counters, as
# profiles/collect_traffic.sh does for the bf16 one.  Run on the GPU box from the repo root.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/pmc32_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc32_$c -- python3 $R/profiles/dominant_kernel.py fp32 > $R/gpurun_out/pmc32_$c.log 2>&1
done
python3 - "$R" <<'PY'
import csv, glob, json, sys
R = sys.argv[1]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = []
    for f in glob.glob("%s/gpurun_out/pmc32_%s/*/*counter_collection.csv" % (R, c)):
        for r in csv.DictReader(open(f)):
            if "k_conv_f32_mfma" in r["Kernel_Name"] and r["Counter_Name"] == c:
                vals.append(float(r["Counter_Value"]))
    out[c] = vals
fetch = sum(out["FETCH_SIZE"]) / max(1, len(out["FETCH_SIZE"]))
write = sum(out["WRITE_SIZE"]) / max(1, len(out["WRITE_SIZE"]))
res = {
    "kernel": "k_conv_f32_mfma<1,8,8,1> conv3d fwd 32->16 3x3x3 @128^3 fp32",
    "launches_sampled": [len(out["FETCH_SIZE"]), len(out["WRITE_SIZE"])],
    "FETCH_SIZE_KB_reported": fetch, "WRITE_SIZE_KB_reported": write,
    "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -> doubled; WRITE_SIZE exact (MI355X_MICROARCH.md, HBM)",
    "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
    "algorithmic_bytes_per_launch": 128 ** 3 * (32 + 16) * 4 + 27 * 32 * 16 * 4,
}
json.dump(res, open(R + "/gpurun_out/dominant_kernel_traffic_fp32.json", "w"), indent=1)
print(json.dumps(res))
PY
