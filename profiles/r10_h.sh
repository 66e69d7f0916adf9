R=$GRAFT_REPO_ROOT
cd $R
bash profiles/ab_cfg.sh - "UNET_TAIL_MAIN=2" "UNET_TAIL_MAIN=1" "UNET_TAIL_MAIN=3" "UNET_TAIL_MAIN=7" "UNET_TAIL_MAIN=2 UNET_POLITE_TAIL_FULL=1" 2>&1 | grep -v amdgpu.ids > gpurun_out/r10h_ab.txt
cat gpurun_out/r10h_ab.txt
bash profiles/profile_step_clean.sh r10h > /dev/null 2>&1
python3 profiles/timeline.py gpurun_out/prof_r10h_clean/runc_kernel_trace.csv 2 400 > gpurun_out/r10h_timeline.txt 2>&1
head -1 gpurun_out/r10h_timeline.txt; tail -24 gpurun_out/r10h_timeline.txt | cut -c1-140
