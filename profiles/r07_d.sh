R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r07d_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r07d_tests.log
tail -4 gpurun_out/r07d_tests.log
grep -h "gradient samples" gpurun_out/r07d_tests.log
bash profiles/ab_env.sh UNET_NO_NORM_BWD_WHOLE 2>&1 | grep -v amdgpu
