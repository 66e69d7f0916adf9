R=$GRAFT_REPO_ROOT
cd $R
bash profiles/ab_cfg.sh - "UNET_PACK_DGRAD_EARLY=1" "UNET_PACK_GRID=512" "UNET_PACK_GRID=1024" "UNET_POLITE_TAIL_FULL=1" 2>&1 | grep -v amdgpu.ids > gpurun_out/r10f_ab.txt
cat gpurun_out/r10f_ab.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r10f_tests.log 2>&1
echo "rc $?" >> gpurun_out/r10f_tests.log
tail -3 gpurun_out/r10f_tests.log
bash profiles/profile_step_clean.sh r10f > /dev/null 2>&1
python3 profiles/timeline.py gpurun_out/prof_r10f_clean/runc_kernel_trace.csv 2 400 > gpurun_out/r10f_timeline.txt 2>&1
head -1 gpurun_out/r10f_timeline.txt; grep pack_batched gpurun_out/r10f_timeline.txt
