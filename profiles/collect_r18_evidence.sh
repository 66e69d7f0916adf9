# round-4 evidence at HEAD, one GPU call:  gpurun --timeout 1200 -- 'bash profiles/collect_r18_evidence.sh [tag]'
set -x
R=$GRAFT_REPO_ROOT
cd $R
T=${1:-r18}
python bench.py --steps 50 --warmup 10 > gpurun_out/${T}_bench_full.json 2> gpurun_out/${T}_bench_full.err
bash profiles/profile_step_clean.sh $T > /dev/null 2>&1
python3 profiles/timeline.py gpurun_out/prof_${T}_clean/runc_kernel_trace.csv 2 400 > gpurun_out/${T}_timeline.txt 2>&1
TOP=400 python profiles/step_profile.py 128 1 5 > gpurun_out/${T}_step_profile_per_op.txt 2>&1
# one-stream trace: the engine's issue order -> the launch sequence DESIGN.md section 4 summarises, and the solo half of the stretch analysis
cd /tmp && export TMPDIR=/tmp
UNET_NO_SIDE_STREAM=1 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_${T}_solo -o runc -- python3 $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-profile --no-kernels --no-cpp-host --batch 0 > $R/gpurun_out/prof_${T}_solo.log 2>&1
cd $R
python3 profiles/launch_sequence.py gpurun_out/prof_${T}_solo/runc_kernel_trace.csv 2 > gpurun_out/${T}_step_launch_sequence.txt 2>&1
python3 profiles/stretch.py gpurun_out/prof_${T}_clean/runc_kernel_trace.csv gpurun_out/prof_${T}_solo/runc_kernel_trace.csv 8 60 > gpurun_out/${T}_stretch.txt 2>&1
bash profiles/collect_step_traffic.sh > gpurun_out/${T}_step_hbm_traffic_per_kernel.txt 2>&1
cp gpurun_out/step_hbm_traffic.json gpurun_out/${T}_step_hbm_traffic.json
bash profiles/collect_traffic.sh > /dev/null 2>&1
bash profiles/collect_traffic.sh wgrad > /dev/null 2>&1
python profiles/bench_wgrad.py > gpurun_out/${T}_bench_wgrad.txt 2>&1
for n in 32 64 128; do python bench.py --size $n --steps 40 --warmup 10 --no-cpu-baseline --no-kernels --no-profile --no-cpp-host --batch 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('size $n  ms_per_step %.4f' % d['ms_per_step'])"; done > gpurun_out/${T}_step_by_size.txt 2>&1
python profiles/host_time.py 2>/dev/null | tail -1 >> gpurun_out/${T}_step_by_size.txt
python profiles/forward_only.py 128 > gpurun_out/${T}_forward_only.json 2>/dev/null
python profiles/soak_train.py > gpurun_out/${T}_soak.txt 2>&1
tail -c 600 gpurun_out/${T}_bench_full.json
