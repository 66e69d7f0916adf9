#!/bin/bash
# Pipe-utilisation counters of the dominant kernel (one rocprofv3 --pmc pass per counter group, --kernel-trace only).
# Run on the GPU box from the repo root:  bash profiles/collect_pipe_counters.sh <tag>
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-pipe}
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"; do
    i=$((i+1))
    rm -rf $R/gpurun_out/pmc_${TAG}_$i
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_$i -- python3 $R/profiles/dominant_kernel.py > $R/gpurun_out/pmc_${TAG}_$i.log 2>&1 || echo "group $i failed: $grp"
done
python3 - "$R" "$TAG" <<'PY'
import csv, glob, sys, collections
R, TAG = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob("%s/gpurun_out/pmc_%s_*/*/*counter_collection.csv" % (R, TAG)):
    for r in csv.DictReader(open(f)):
        if "k_mfma_conv_z" in r["Kernel_Name"] or "k_mfma_conv_p" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print("%-28s %16.0f  (n=%d)" % (k, sum(acc[k]) / len(acc[k]), len(acc[k])))
PY
