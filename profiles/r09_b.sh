R=$GRAFT_REPO_ROOT
cd $R
for v in 300 600 1100 2100; do bash profiles/ab_envval.sh UNET_SMALL_BELOW $v 2>&1 | grep -v amdgpu; done
