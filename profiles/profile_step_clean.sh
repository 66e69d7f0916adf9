#!/bin/bash
# rocprofv3 kernel trace of the train step ONLY (no micro-benchmarks, no per-op profile pass, no CPU baseline):
#   gpurun -- 'bash profiles/profile_step_clean.sh r05'
# leaves gpurun_out/<tag>_step_per_kernel_per_grid.txt (per (kernel, grid) time per step) and gpurun_out/<tag>_step_kernel_stats.csv
set -e
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_clean -o runc -- python3 $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-profile --no-kernels --no-cpp-host --batch 0 > $R/gpurun_out/prof_${TAG}_clean.log 2>&1
python3 $R/profiles/summarize.py $R/gpurun_out/prof_${TAG}_clean/runc_kernel_trace.csv 20 90 > $R/gpurun_out/${TAG}_step_per_kernel_per_grid.txt
cp $R/gpurun_out/prof_${TAG}_clean/runc_kernel_stats.csv $R/gpurun_out/${TAG}_step_kernel_stats.csv
grep -h '^{' $R/gpurun_out/prof_${TAG}_clean.log | tail -1
