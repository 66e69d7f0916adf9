#!/bin/bash
# Pipe-utilisation counters AND HBM traffic of one kernel: one rocprofv3 --pmc pass per counter group (--kernel-trace only next to
# --pmc; FETCH_SIZE and WRITE_SIZE in passes of their own, MI355X_MICROARCH.md "rocprofv3 PMC slots").
# Run on the GPU box from the repo root:
#     bash profiles/collect_counters.sh <tag> <kernel-name-substring> <script.py> [script args...]
# Prints the per-launch averages and writes gpurun_out/<tag>_counters.txt (copy it into profiles/ to keep it).
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; MATCH=$2; SCRIPT=$(cd "$(dirname "$3")" && pwd)/$(basename "$3"); shift 3
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rm -rf $R/gpurun_out/pmc_${TAG}_$i
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_$i -- python3 "$SCRIPT" "$@" > $R/gpurun_out/pmc_${TAG}_$i.log 2>&1 || echo "group $i failed: $grp"
done
python3 - "$R" "$TAG" "$MATCH" "$SCRIPT $*" <<'PY' | tee $R/gpurun_out/${TAG}_counters.txt
import csv, glob, sys, collections
R, TAG, MATCH, CMD = sys.argv[1:5]
acc = collections.defaultdict(list)
dur = []
for f in glob.glob("%s/gpurun_out/pmc_%s_*/*counter_collection.csv" % (R, TAG)) + glob.glob("%s/gpurun_out/pmc_%s_*/*/*counter_collection.csv" % (R, TAG)):
    for r in csv.DictReader(open(f)):
        if MATCH in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("%s/gpurun_out/pmc_%s_1/*kernel_trace.csv" % (R, TAG)) + glob.glob("%s/gpurun_out/pmc_%s_1/*/*kernel_trace.csv" % (R, TAG)):
    for r in csv.DictReader(open(f)):
        if MATCH in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("# kernel '%s' of: %s" % (MATCH, CMD))
if dur:
    print("%-28s %16.1f  us (n=%d, under counter collection)" % ("duration", sum(dur) / len(dur), len(dur)))
for k in sorted(acc):
    print("%-28s %16.0f  (n=%d)" % (k, sum(acc[k]) / len(acc[k]), len(acc[k])))
if "FETCH_SIZE" in acc and "WRITE_SIZE" in acc:
    f, w = sum(acc["FETCH_SIZE"]) / len(acc["FETCH_SIZE"]), sum(acc["WRITE_SIZE"]) / len(acc["WRITE_SIZE"])
    print("%-28s %16.0f  bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB (gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads)" % ("hbm_bytes", (2 * f + w) * 1024))
PY
