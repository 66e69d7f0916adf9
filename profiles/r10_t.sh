R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r10t_tests.log 2>&1
echo "rc $?" >> gpurun_out/r10t_tests.log
tail -6 gpurun_out/r10t_tests.log
bash profiles/ab_cfg.sh - "UNET_NO_DGRAD_BNSTATS_SMALL=1" 2>&1 | grep -v amdgpu.ids > gpurun_out/r10t_ab.txt
cat gpurun_out/r10t_ab.txt
