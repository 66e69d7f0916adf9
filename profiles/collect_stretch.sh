# both kernel traces of the stretch analysis in ONE GPU call (gpurun_out/ does not travel to the box):  bash profiles/collect_stretch.sh [tag]
R=$GRAFT_REPO_ROOT
cd $R
T=${1:-r18}
bash profiles/profile_step_clean.sh $T > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
UNET_NO_SIDE_STREAM=1 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_${T}_solo -o runc -- python3 $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-profile --no-kernels --no-cpp-host --batch 0 > $R/gpurun_out/prof_${T}_solo.log 2>&1
cd $R
python3 profiles/stretch.py gpurun_out/prof_${T}_clean/runc_kernel_trace.csv gpurun_out/prof_${T}_solo/runc_kernel_trace.csv 8 60 > gpurun_out/${T}_stretch.txt 2>&1
python3 profiles/timeline.py gpurun_out/prof_${T}_clean/runc_kernel_trace.csv 2 400 > gpurun_out/${T}_timeline.txt 2>&1
python3 profiles/launch_sequence.py gpurun_out/prof_${T}_solo/runc_kernel_trace.csv 2 > gpurun_out/${T}_step_launch_sequence.txt 2>&1
head -3 gpurun_out/${T}_stretch.txt; tail -3 gpurun_out/${T}_timeline.txt
