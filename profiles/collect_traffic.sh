#!/bin/bash
# HBM traffic of the dominant kernel from PMC counters: two separate rocprofv3 passes (FETCH_SIZE needs 3 of the 4 TCC
# slots, WRITE_SIZE 2), --kernel-trace only next to --pmc.  Writes profiles/dominant_kernel_traffic.json.
# Run on the GPU box from the repo root:  bash profiles/collect_traffic.sh
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/pmc_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/profiles/dominant_kernel.py > $R/gpurun_out/pmc_$c.log 2>&1
done
mkdir -p $R/gpurun_out; python3 - "$R" <<'PY'
import csv, glob, json, sys
R = sys.argv[1]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = []
    for f in glob.glob("%s/gpurun_out/pmc_%s/*/*counter_collection.csv" % (R, c)):
        for r in csv.DictReader(open(f)):
            if ("k_mfma_conv_z" in r["Kernel_Name"] or "k_mfma_conv_p" in r["Kernel_Name"]) and r["Counter_Name"] == c:
                vals.append(float(r["Counter_Value"]))
    out[c] = vals
fetch = sum(out["FETCH_SIZE"]) / max(1, len(out["FETCH_SIZE"]))   # KB per launch as reported
write = sum(out["WRITE_SIZE"]) / max(1, len(out["WRITE_SIZE"]))
res = {
    "kernel": "k_mfma_conv_z (sliding-window MFMA conv) conv3d fwd 32->16 3x3x3 @128^3 bf16 + stats epilogue",
    "launches_sampled": [len(out["FETCH_SIZE"]), len(out["WRITE_SIZE"])],
    "FETCH_SIZE_KB_reported": fetch, "WRITE_SIZE_KB_reported": write,
    "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -> doubled; WRITE_SIZE exact (MI355X_MICROARCH.md, HBM)",
    "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
    "algorithmic_bytes_per_launch": 128 ** 3 * (32 + 16) * 2 + 27 * 32 * 16 * 2,
}
json.dump(res, open(R + "/profiles/dominant_kernel_traffic.json", "w"), indent=1)
json.dump(res, open(R + "/gpurun_out/dominant_kernel_traffic.json", "w"), indent=1)   # gpurun merges only gpurun_out/ back: copy it to profiles/
print(json.dumps(res))
PY
