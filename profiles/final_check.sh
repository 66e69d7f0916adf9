# end-of-round check on one box: the GPU test suite, the smoke entry, and the default bench line three times
R=$GRAFT_REPO_ROOT
cd $R
T=${1:-final}
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method=thread > gpurun_out/${T}_tests.log 2>&1
echo "rc $?" >> gpurun_out/${T}_tests.log
tail -3 gpurun_out/${T}_tests.log
python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' > gpurun_out/${T}_smoke.log 2>&1
tail -1 gpurun_out/${T}_smoke.log
for i in 1 2 3; do python bench.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench.py default run $i: ms_per_step %.4f  value %.0f voxels/s  roofline.frac %.3f  batch8 ms/sample %.3f' % (d['ms_per_step'], d['value'], d['roofline']['frac'], d['batch8']['ms_per_sample']))"; done > gpurun_out/${T}_bench3.txt 2>&1
cat gpurun_out/${T}_bench3.txt
