R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r07j_tests.log 2>&1
echo "rc $?" >> gpurun_out/r07j_tests.log
tail -3 gpurun_out/r07j_tests.log
python profiles/forward_only.py 128 2>/dev/null | tail -1
UNET_NO_F32_STATS_EPILOGUE=1 UNET_FWD_REPACK=1 python profiles/forward_only.py 128 2>/dev/null | tail -1
