#!/bin/bash
# HBM traffic of one op's kernels from PMC counters: two separate rocprofv3 passes (FETCH_SIZE needs 3 of the 4 TCC slots, WRITE_SIZE 2),
# --kernel-trace only next to --pmc.  Writes profiles/<name>.json (read by bench.py as "recorded" traffic).
#   bash profiles/collect_traffic.sh                       -> dominant_kernel_traffic.json  (conv fwd 32->16 @128^3, profiles/dominant_kernel.py)
#   bash profiles/collect_traffic.sh wgrad                 -> wgrad_kernel_traffic.json     (wgrad 32->16 @128^3 + its slab reduce, profiles/wgrad_kernel.py)
# Run on the GPU box from the repo root.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp UNET_OP_POLITE=1     # the weight gradient as the train step launches it (polite: engine.cpp, choose_polite)
WHICH=${1:-conv}
if [ "$WHICH" = wgrad ]; then NAME=wgrad_kernel_traffic; SCRIPT="$R/profiles/wgrad_kernel.py 32 16 128"; else NAME=dominant_kernel_traffic; SCRIPT="$R/profiles/dominant_kernel.py"; fi
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/pmc_${WHICH}_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${WHICH}_$c -- python3 $SCRIPT > $R/gpurun_out/pmc_${WHICH}_$c.log 2>&1
done
mkdir -p $R/gpurun_out; python3 - "$R" "$WHICH" "$NAME" <<'PY'
import csv, glob, json, sys, collections
R, WHICH, NAME = sys.argv[1:4]
match = ("k_mfma_wgrad_z", "k_mfma_wgrad<", "k_mfma_wgrad_reduce", "k_wgrad_reduce") if WHICH == "wgrad" else ("k_mfma_conv_z", "k_mfma_conv_p")
per, launches = {}, {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc, n = collections.Counter(), collections.Counter()
    for f in glob.glob("%s/gpurun_out/pmc_%s_%s/*/*counter_collection.csv" % (R, WHICH, c)):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and any(m in r["Kernel_Name"] for m in match):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("unet::", "")
                acc[k] += float(r["Counter_Value"]); n[k] += 1
    per[c], launches[c] = acc, n
kernels = {}
for k in set(per["FETCH_SIZE"]) | set(per["WRITE_SIZE"]):
    f = per["FETCH_SIZE"][k] / max(1, launches["FETCH_SIZE"][k]); w = per["WRITE_SIZE"][k] / max(1, launches["WRITE_SIZE"][k])
    kernels[k] = {"FETCH_SIZE_KB_reported": f, "WRITE_SIZE_KB_reported": w, "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
                  "launches_sampled": [launches["FETCH_SIZE"][k], launches["WRITE_SIZE"][k]]}
alg = 128 ** 3 * (32 + 16) * 2 + 27 * 32 * 16 * (4 if WHICH == "wgrad" else 2)
res = {
    "kernel": ("conv3d wgrad 32->16 3x3x3 @128^3 bf16: " if WHICH == "wgrad" else "conv3d fwd 32->16 3x3x3 @128^3 bf16 + stats epilogue: ") + " + ".join(sorted(kernels)),
    "kernels": kernels,
    "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -> doubled; WRITE_SIZE exact (MI355X_MICROARCH.md, HBM)",
    "hbm_bytes_per_launch": sum(v["hbm_bytes_per_launch"] for v in kernels.values()),
    "algorithmic_bytes_per_launch": alg,
}
json.dump(res, open(R + "/profiles/%s.json" % NAME, "w"), indent=1)
json.dump(res, open(R + "/gpurun_out/%s.json" % NAME, "w"), indent=1)   # gpurun merges only gpurun_out/ back: copy it to profiles/
print(json.dumps(res))
PY
