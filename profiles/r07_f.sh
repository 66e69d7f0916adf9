R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "micro or lane or resume or golden or two_rank_gpu_one_step" > gpurun_out/r07f_tests.log 2>&1
tail -3 gpurun_out/r07f_tests.log
for v in 1 2; do
export UNET_DEBUG_WHOLE=$v
bash profiles/profile_step_clean.sh r07f$v > /dev/null 2>&1
python3 profiles/timeline.py gpurun_out/prof_r07f${v}_clean/runc_kernel_trace.csv 2 400 > gpurun_out/r07f${v}_timeline.txt 2>&1
echo "UNET_DEBUG_WHOLE=$v"; grep whole8 gpurun_out/r07f${v}_timeline.txt | cut -c1-40,60-130
done
unset UNET_DEBUG_WHOLE
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernels --no-profile > gpurun_out/r07f_bench_if2.json 2>/dev/null
UNET_MICRO_IN_FLIGHT=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernels --no-profile > gpurun_out/r07f_bench_if1.json 2>/dev/null
python -c "
import json
for f in ('if2','if1'):
    d=json.loads(open('gpurun_out/r07f_bench_%s.json'%f).read().strip().splitlines()[-1]); print(f, round(d['ms_per_step'],4), d['batch8'])
"
