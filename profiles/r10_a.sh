R=$GRAFT_REPO_ROOT
cd $R
bash profiles/ab_cfg.sh - "UNET_NO_STATS_DIRECT=1" "UNET_SIDE_POLITE=83000" "UNET_SIDE_POLITE=83000 UNET_WZ_P11=100000000" "UNET_WZ_P11=100000000" "UNET_NO_SIDE_STREAM=1" > gpurun_out/r10a_ab.txt 2>&1
cat gpurun_out/r10a_ab.txt
bash profiles/profile_step_clean.sh r10a > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
UNET_NO_SIDE_STREAM=1 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_r10a_solo -o runc -- python3 $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-profile --no-kernels --batch 0 > $R/gpurun_out/prof_r10a_solo.log 2>&1
cd $R
python3 profiles/stretch.py gpurun_out/prof_r10a_clean/runc_kernel_trace.csv gpurun_out/prof_r10a_solo/runc_kernel_trace.csv 8 60 > gpurun_out/r10a_stretch.txt 2>&1
head -5 gpurun_out/r10a_stretch.txt
