set -x
R=$GRAFT_REPO_ROOT
cd $R
python bench.py --steps 50 --warmup 10 > gpurun_out/r04_bench_full.json 2> gpurun_out/r04_bench_full.err
bash profiles/profile_bench.sh r04 > /dev/null 2>&1
python3 profiles/summarize.py gpurun_out/prof_r04/runc_kernel_trace.csv 10 70 > gpurun_out/r04_bench_per_kernel_per_grid.txt 2>&1
cp gpurun_out/prof_r04/runc_kernel_stats.csv gpurun_out/r04_bench_kernel_stats.csv
grep -h '^{' gpurun_out/prof_r04.log | tail -1 > gpurun_out/r04_bench_profiled.json
bash profiles/collect_counters.sh r04wz32 k_mfma_wgrad_z profiles/wgrad_kernel.py 32 16 128 > /dev/null 2>&1
bash profiles/collect_counters.sh r04wz16 k_mfma_wgrad_z profiles/wgrad_kernel.py 16 16 128 > /dev/null 2>&1
TOP=400 python profiles/step_profile.py 128 1 5 > gpurun_out/r04_step_profile_per_op.txt 2>&1
python profiles/bw_probe.py > gpurun_out/r04_bw_probe.txt 2>&1
python profiles/bench_wgrad.py > gpurun_out/r04_bench_wgrad.txt 2>&1
tail -c 1500 gpurun_out/r04_bench_full.json
