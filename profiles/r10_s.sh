R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "conv" > gpurun_out/r10s_tests1.log 2>&1
echo "rc $?" >> gpurun_out/r10s_tests1.log
tail -3 gpurun_out/r10s_tests1.log
bash profiles/ab_cfg.sh - "UNET_CONVT_DGRAD_CK16=1" 2>&1 | grep -v amdgpu.ids > gpurun_out/r10s_ab.txt
cat gpurun_out/r10s_ab.txt
bash profiles/profile_step_clean.sh r10s > /dev/null 2>&1
grep -n "k_mfma_conv_p<2, 2, 0" gpurun_out/r10s_step_per_kernel_per_grid.txt
