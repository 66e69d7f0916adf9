# everything DESIGN.md / the bench line cite for this round, in one GPU call:  gpurun -- 'bash profiles/collect_r06_evidence.sh'
set -x
R=$GRAFT_REPO_ROOT
cd $R
python bench.py --steps 50 --warmup 10 > gpurun_out/r06_bench_full.json 2> gpurun_out/r06_bench_full.err
bash profiles/profile_step_clean.sh r06 > /dev/null 2>&1
bash profiles/profile_bench.sh r06 > /dev/null 2>&1
python3 profiles/summarize.py gpurun_out/prof_r06/runc_kernel_trace.csv 10 70 > gpurun_out/r06_bench_per_kernel_per_grid.txt 2>&1
cp gpurun_out/prof_r06/runc_kernel_stats.csv gpurun_out/r06_bench_kernel_stats.csv
grep -h '^{' gpurun_out/prof_r06.log | tail -1 > gpurun_out/r06_bench_profiled.json
bash profiles/collect_step_traffic.sh > gpurun_out/r06_step_hbm_traffic_per_kernel.txt 2>&1
bash profiles/collect_traffic.sh > /dev/null 2>&1
bash profiles/collect_traffic.sh wgrad > /dev/null 2>&1
bash profiles/collect_counters.sh r06_conv_z16_16to16_128 k_mfma_conv_z16 profiles/dgrad_kernel.py 16 16 128 > /dev/null 2>&1
bash profiles/collect_counters.sh r06_conv_z32_32to16_128 k_mfma_conv_z32 profiles/dominant_kernel.py > /dev/null 2>&1
bash profiles/collect_counters.sh r06_wgrad_z_32to16_128 k_mfma_wgrad_z profiles/wgrad_kernel.py 32 16 128 > /dev/null 2>&1
bash profiles/collect_counters.sh r06_wgrad_z_16to16_128 k_mfma_wgrad_z profiles/wgrad_kernel.py 16 16 128 > /dev/null 2>&1
TOP=400 python profiles/step_profile.py 128 1 5 > gpurun_out/r06_step_profile_per_op.txt 2>&1
python profiles/bw_probe.py > gpurun_out/r06_bw_probe.txt 2>&1
python profiles/bench_wgrad.py > gpurun_out/r06_bench_wgrad.txt 2>&1
(unet-studio_amd/csrc/build/elem_bench 128 16 && unet-studio_amd/csrc/build/elem_bench 64 32 && unet-studio_amd/csrc/build/elem_bench 32 64) > gpurun_out/r06_elem_bench.txt 2>&1
tail -c 1500 gpurun_out/r06_bench_full.json
