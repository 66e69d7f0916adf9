R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "full_size or golden or buckets or two_rank_gpu_one_step" > gpurun_out/r07i_tests.log 2>&1
tail -2 gpurun_out/r07i_tests.log
bash profiles/ab_envval.sh UNET_FORK_EVERY 1 2>&1 | grep -v amdgpu
for n in 32 64; do python bench.py --size $n --steps 40 --warmup 10 --no-cpu-baseline --no-kernels --no-profile --batch 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('size $n ms_per_step', round(d['ms_per_step'],4))"; done
python profiles/host_time.py 2>/dev/null | tail -1
bash profiles/profile_step_clean.sh r07i > /dev/null 2>&1
python3 profiles/timeline.py gpurun_out/prof_r07i_clean/runc_kernel_trace.csv 2 400 > gpurun_out/r07i_timeline.txt 2>&1
head -1 gpurun_out/r07i_timeline.txt; tail -12 gpurun_out/r07i_timeline.txt | cut -c1-130
