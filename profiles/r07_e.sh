R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s -k "golden" > gpurun_out/r07e_tests.log 2>&1
grep -h "gradient samples\|passed\|failed" gpurun_out/r07e_tests.log
bash profiles/profile_step_clean.sh r07e > /dev/null 2>&1
python3 profiles/timeline.py gpurun_out/prof_r07e_clean/runc_kernel_trace.csv 2 400 > gpurun_out/r07e_timeline.txt 2>&1
grep -c . gpurun_out/r07e_timeline.txt
