# everything DESIGN.md / the bench line cite for round 3, in one GPU call:  gpurun -- 'bash profiles/collect_evidence.sh [tag]'
set -x
R=$GRAFT_REPO_ROOT
cd $R
T=${1:-r11}
python bench.py --steps 50 --warmup 10 > gpurun_out/${T}_bench_full.json 2> gpurun_out/${T}_bench_full.err
bash profiles/profile_step_clean.sh $T > /dev/null 2>&1
python3 profiles/timeline.py gpurun_out/prof_${T}_clean/runc_kernel_trace.csv 2 400 > gpurun_out/${T}_timeline.txt 2>&1
bash profiles/profile_bench.sh $T > /dev/null 2>&1
python3 profiles/summarize.py gpurun_out/prof_$T/runc_kernel_trace.csv 0 70 > gpurun_out/${T}_bench_per_kernel_per_grid.txt 2>&1
cp gpurun_out/prof_$T/runc_kernel_stats.csv gpurun_out/${T}_bench_kernel_stats.csv
grep -h '^{' gpurun_out/prof_$T.log | tail -1 > gpurun_out/${T}_bench_profiled.json
bash profiles/collect_step_traffic.sh > gpurun_out/${T}_step_hbm_traffic_per_kernel.txt 2>&1
bash profiles/collect_traffic.sh > /dev/null 2>&1
bash profiles/collect_traffic.sh wgrad > /dev/null 2>&1
TOP=400 python profiles/step_profile.py 128 1 5 > gpurun_out/${T}_step_profile_per_op.txt 2>&1
python profiles/bench_wgrad.py > gpurun_out/${T}_bench_wgrad.txt 2>&1
# the output-stationary wgrad that adds into the gradient (256->256 @8^3) and the stride-2 / conv_trans weight gradients: pipe + traffic counters
bash profiles/collect_counters.sh ${T}_wgrad_direct_256to256_8 "k_mfma_wgrad<1, 3, 1, 8, 8, 8, 1, 1>" profiles/wgrad_kernel.py 256 256 8 > /dev/null 2>&1
bash profiles/collect_counters.sh ${T}_wgrad_s2_16to32_128 "k_mfma_wgrad<2, 3, 1" profiles/wgrad_kernel.py 16 32 128 2 > /dev/null 2>&1
# where the step's time is: fixed (launch chain) vs size-dependent -- the same step at 32^3, 64^3, 128^3
for n in 32 64 128; do python bench.py --size $n --steps 40 --warmup 10 --no-cpu-baseline --no-kernels --no-profile --batch 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('size $n  ms_per_step %.4f' % d['ms_per_step'])"; done > gpurun_out/${T}_step_by_size.txt 2>&1
python profiles/host_time.py 2>/dev/null | tail -1 >> gpurun_out/${T}_step_by_size.txt
# batch 8 with one and with two micro-steps in flight
for f in 1 2; do UNET_MICRO_IN_FLIGHT=$f python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-kernels --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('in_flight $f', json.dumps(d['batch8']))"; done > gpurun_out/${T}_batch8_in_flight.txt 2>&1
python profiles/forward_only.py 128 > gpurun_out/${T}_forward_only.json 2>/dev/null
python profiles/bench_evaluate.py > gpurun_out/${T}_bench_evaluate.json 2>/dev/null
tail -c 1500 gpurun_out/${T}_bench_full.json
# soak: 300 steps, loss and memory every 25 (profiles/soak_train.py)
python profiles/soak_train.py > gpurun_out/${T}_soak.txt 2>&1
# how much longer the caller's stream's kernels take beside the side stream than alone (profiles/stretch.py)
cd /tmp && export TMPDIR=/tmp
UNET_NO_SIDE_STREAM=1 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_${T}_solo -o runc -- python3 $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-profile --no-kernels --batch 0 > $R/gpurun_out/prof_${T}_solo.log 2>&1
cd $R
python3 profiles/stretch.py gpurun_out/prof_${T}_clean/runc_kernel_trace.csv gpurun_out/prof_${T}_solo/runc_kernel_trace.csv 8 60 > gpurun_out/${T}_stretch.txt 2>&1
