R=$GRAFT_REPO_ROOT
cd $R
L="UNET_PACK_DGRAD_LATE=1"
bash profiles/ab_cfg.sh - "$L UNET_PACK_GRID_LATE=256" "$L UNET_PACK_GRID_LATE=128" "$L UNET_PACK_GRID_LATE=512" "$L UNET_PACK_GRID_LATE=64" 2>&1 | grep -v amdgpu.ids > gpurun_out/r10j_ab.txt
cat gpurun_out/r10j_ab.txt
