R=$GRAFT_REPO_ROOT
cd $R
bash profiles/ab_cfg.sh - "UNET_SIDE_POLITE=0" "UNET_POLITE_ALL=1" "UNET_WZ_BLOCKSP=192" "UNET_WZ_BLOCKSP=512" "UNET_WGRAD_HOLD=0" 2>&1 | grep -v amdgpu.ids > gpurun_out/r10e_ab.txt
cat gpurun_out/r10e_ab.txt
bash profiles/profile_step_clean.sh r10e > /dev/null 2>&1
python3 profiles/timeline.py gpurun_out/prof_r10e_clean/runc_kernel_trace.csv 2 400 > gpurun_out/r10e_timeline.txt 2>&1
cd /tmp && export TMPDIR=/tmp
UNET_NO_SIDE_STREAM=1 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_r10e_solo -o runc -- python3 $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-profile --no-kernels --batch 0 > $R/gpurun_out/prof_r10e_solo.log 2>&1
cd $R
python3 profiles/stretch.py gpurun_out/prof_r10e_clean/runc_kernel_trace.csv gpurun_out/prof_r10e_solo/runc_kernel_trace.csv 8 60 > gpurun_out/r10e_stretch.txt 2>&1
head -3 gpurun_out/r10e_stretch.txt | cut -c1-250
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r10e_tests.log 2>&1
echo "rc $?" >> gpurun_out/r10e_tests.log
tail -3 gpurun_out/r10e_tests.log
