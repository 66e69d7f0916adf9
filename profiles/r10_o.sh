R=$GRAFT_REPO_ROOT
cd $R
bash profiles/ab_cfg.sh - "UNET_LOSS_FORK_PER_LEVEL=1" "UNET_LOSS_FORK_PER_LEVEL=1 UNET_HEAD_VIEW=0" "UNET_HEAD_VIEW=0" 2>&1 | grep -v amdgpu.ids > gpurun_out/r10o_ab.txt
cat gpurun_out/r10o_ab.txt
UNET_LOSS_FORK_PER_LEVEL=1 bash profiles/profile_step_clean.sh r10o > /dev/null 2>&1
python3 profiles/timeline.py gpurun_out/prof_r10o_clean/runc_kernel_trace.csv 2 400 > gpurun_out/r10o_timeline.txt 2>&1
sed -n 40,100p gpurun_out/r10o_timeline.txt | cut -c1-130
