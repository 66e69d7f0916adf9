#!/bin/bash
# rocprofv3 kernel statistics of the fp32 train step (bench.py --dtype fp32), from the repo root on the GPU box
set -e
ROOT=$(pwd)
mkdir -p gpurun_out/prof_f32step
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_f32step -o st -- python3 $ROOT/bench.py --dtype fp32 --steps 2 --warmup 1 --no-cpu-baseline > $ROOT/gpurun_out/prof_f32step/stdout.log 2>&1
cd $ROOT
cp $(find gpurun_out/prof_f32step -name '*kernel_stats.csv' | head -1) gpurun_out/fp32_step_kernel_stats.csv
