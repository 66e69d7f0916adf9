# baseline of round 3 on this round's box: GPU tests, bench line, clean kernel trace + one step's time line, per-op profile
set -x
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests -m gpu -x -q > gpurun_out/r07_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r07_tests.log
tail -3 gpurun_out/r07_tests.log
python bench.py --steps 30 --warmup 10 > gpurun_out/r07_bench_full.json 2> gpurun_out/r07_bench_full.err
bash profiles/profile_step_clean.sh r07 > /dev/null 2>&1
python3 profiles/timeline.py gpurun_out/prof_r07_clean/runc_kernel_trace.csv 2 400 > gpurun_out/r07_timeline.txt 2>&1
TOP=400 python profiles/step_profile.py 128 1 5 > gpurun_out/r07_step_profile_per_op.txt 2>&1
tail -c 1800 gpurun_out/r07_bench_full.json
