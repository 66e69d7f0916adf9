R=$GRAFT_REPO_ROOT
cd $R
bash profiles/ab_cfg.sh - "UNET_PACK_AFTER_UPDATE=1" 2>&1 | grep -v amdgpu.ids > gpurun_out/r10k_ab.txt
cat gpurun_out/r10k_ab.txt
