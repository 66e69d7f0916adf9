R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "first_conv or bnstats or golden or statistics_in_the_dgrad" > gpurun_out/r10g_tests1.log 2>&1
echo "rc $?" >> gpurun_out/r10g_tests1.log
tail -5 gpurun_out/r10g_tests1.log
bash profiles/ab_cfg.sh - "UNET_NO_FIRST_WGRAD_FUSE=1" 2>&1 | grep -v amdgpu.ids > gpurun_out/r10g_ab.txt
cat gpurun_out/r10g_ab.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r10g_tests.log 2>&1
echo "rc $?" >> gpurun_out/r10g_tests.log
tail -3 gpurun_out/r10g_tests.log
