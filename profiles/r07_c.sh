R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_cpp_host.py tests/test_gpu_parity.py -m gpu -x -q -k "cpp_drop_in or conv3d_ops or convt_ops or resume or micro_steps or golden or network" > gpurun_out/r07c_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r07c_tests.log
tail -4 gpurun_out/r07c_tests.log
python profiles/bench_wgrad.py > gpurun_out/r07c_bench_wgrad.txt 2>&1
grep -v amdgpu gpurun_out/r07c_bench_wgrad.txt
bash profiles/ab_env.sh UNET_NO_WGRAD_DIRECT 2>&1 | grep -v amdgpu
python bench.py --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/r07c_bench.json 2> gpurun_out/r07c_bench.err
tail -c 3000 gpurun_out/r07c_bench.json
