R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r08b_tests.log 2>&1
echo "rc $?" >> gpurun_out/r08b_tests.log
tail -3 gpurun_out/r08b_tests.log
python profiles/forward_only.py 128 2>/dev/null | tail -1
python profiles/forward_only.py 128 2>/dev/null | tail -1
UNET_FWD_REPACK=1 python profiles/forward_only.py 128 2>/dev/null | tail -1
python profiles/bench_evaluate.py 2>/dev/null | tail -1
