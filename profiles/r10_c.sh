R=$GRAFT_REPO_ROOT
cd $R
timeout -k 5 120 profiles/tools/bin/cu_mask_probe > gpurun_out/r10c_cu_mask_probe.txt 2>&1
cat gpurun_out/r10c_cu_mask_probe.txt
P="UNET_SIDE_POLITE=83000"
bash profiles/ab_cfg.sh - "$P UNET_WZ_P11=100000000 UNET_WZ_BLOCKS=256" "$P UNET_WZ_BLOCKS=256 UNET_WZ_BLOCKS8=128" "$P UNET_WZ_BLOCKS=256 UNET_WZ_BLOCKS8=192" "$P UNET_WZ_P11=100000000 UNET_WZ_BLOCKS=192" "$P UNET_WZ_P11=100000000 UNET_WZ_BLOCKS=128" "$P UNET_WZ_P11=100000000 UNET_WZ_BLOCKS=256 UNET_WGRAD_HOLD=0" "UNET_SIDE_POLITE=120000 UNET_WZ_P11=100000000 UNET_WZ_BLOCKS=256" 2>&1 | grep -v amdgpu.ids > gpurun_out/r10c_ab.txt
cat gpurun_out/r10c_ab.txt
