#!/bin/bash
# rocprofv3 kernel trace of the default bench.py workload (run on the GPU box through gpurun):
#   gpurun -- 'bash profiles/profile_bench.sh r01x'
# leaves gpurun_out/prof_<tag>/ (kernel_trace + kernel_stats CSVs) and gpurun_out/prof_<tag>.log; copy what is to be
# judged into profiles/ with profiles/summarize.py.
set -e
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o runc -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --batch 0 > $R/gpurun_out/prof_$TAG.log 2>&1
grep -h '^{' $R/gpurun_out/prof_$TAG.log | tail -1
