R=$GRAFT_REPO_ROOT
cd $R
bash profiles/ab_cfg.sh - "UNET_NO_FIRST_WGRAD_FUSE=1" "UNET_TAIL_MAIN=0" "UNET_WGRAD_FIRST_BLOCKS=1024" 2>&1 | grep -v amdgpu.ids > gpurun_out/r10i_ab.txt
cat gpurun_out/r10i_ab.txt
bash profiles/profile_step_clean.sh r10i > /dev/null 2>&1
python3 profiles/timeline.py gpurun_out/prof_r10i_clean/runc_kernel_trace.csv 2 400 > gpurun_out/r10i_timeline.txt 2>&1
cd /tmp && export TMPDIR=/tmp
UNET_NO_SIDE_STREAM=1 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_r10i_solo -o runc -- python3 $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-profile --no-kernels --batch 0 > $R/gpurun_out/prof_r10i_solo.log 2>&1
cd $R
python3 profiles/stretch.py gpurun_out/prof_r10i_clean/runc_kernel_trace.csv gpurun_out/prof_r10i_solo/runc_kernel_trace.csv 8 60 > gpurun_out/r10i_stretch.txt 2>&1
head -1 gpurun_out/r10i_timeline.txt; tail -24 gpurun_out/r10i_timeline.txt | cut -c1-140
head -2 gpurun_out/r10i_stretch.txt | cut -c1-250
