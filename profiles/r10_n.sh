R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r10n_tests.log 2>&1
echo "rc $?" >> gpurun_out/r10n_tests.log
tail -4 gpurun_out/r10n_tests.log
bash profiles/ab_cfg.sh - "UNET_HEAD_VIEW=0" 2>&1 | grep -v amdgpu.ids > gpurun_out/r10n_ab.txt
cat gpurun_out/r10n_ab.txt
