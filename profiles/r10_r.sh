R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "conv" > gpurun_out/r10r_tests1.log 2>&1
echo "rc $?" >> gpurun_out/r10r_tests1.log
tail -3 gpurun_out/r10r_tests1.log
bash profiles/ab_cfg.sh - "UNET_SC_NT=4" "UNET_SC_NT=2" "UNET_SC_NT=1" 2>&1 | grep -v amdgpu.ids > gpurun_out/r10r_ab.txt
cat gpurun_out/r10r_ab.txt
bash profiles/profile_step_clean.sh r10r > /dev/null 2>&1
grep -n "true, [48], false" gpurun_out/r10r_step_per_kernel_per_grid.txt
