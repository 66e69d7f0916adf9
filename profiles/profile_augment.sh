#!/bin/bash
# rocprofv3 kernel statistics of the augmentation bench (run on the GPU box from the repo root).
set -e
ROOT=$(pwd)
mkdir -p gpurun_out/prof_aug
python3 profiles/bench_augment.py > gpurun_out/augment_bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_aug -o aug -- python3 $ROOT/profiles/bench_augment.py --iters 5 > $ROOT/gpurun_out/prof_aug/stdout.log 2>&1
cd $ROOT
f=$(find gpurun_out/prof_aug -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/augment_kernel_stats.csv
head -12 gpurun_out/augment_kernel_stats.csv
cat gpurun_out/augment_bench.json
