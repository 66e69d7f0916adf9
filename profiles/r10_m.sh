R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r10m_tests.log 2>&1
echo "rc $?" >> gpurun_out/r10m_tests.log
tail -4 gpurun_out/r10m_tests.log
