#!/bin/bash
# rocprofv3 kernel statistics of the forward-only run (fp32 + bf16), from the repo root on the GPU box
set -e
ROOT=$(pwd)
mkdir -p gpurun_out/prof_fwd
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_fwd -o fwd -- python3 $ROOT/profiles/forward_only.py > $ROOT/gpurun_out/prof_fwd/stdout.log 2>&1
cd $ROOT
cp $(find gpurun_out/prof_fwd -name '*kernel_stats.csv' | head -1) gpurun_out/forward_kernel_stats.csv
head -14 gpurun_out/forward_kernel_stats.csv | cut -c1-200
