R=$GRAFT_REPO_ROOT
cd $R
for v in 1024 1536 2048; do bash profiles/ab_envval.sh UNET_CONV_FIRST_BLOCKS $v 2>&1 | grep -v amdgpu; done
for v in 1024 2048; do bash profiles/ab_envval.sh UNET_WGRAD_FIRST_BLOCKS $v 2>&1 | grep -v amdgpu; done
