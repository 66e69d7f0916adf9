R=$GRAFT_REPO_ROOT
cd $R
P="UNET_SIDE_POLITE=83000 UNET_WZ_P11=100000000"
bash profiles/ab_cfg.sh - "$P" "$P UNET_PACK_GRID=256" "$P UNET_PACK_GRID=512" "UNET_PACK_GRID=256" "$P UNET_WZ_BLOCKS=256" "$P UNET_FORK_EVERY=1" "$P UNET_WZ_FLUSH=1" 2>&1 | grep -v amdgpu.ids > gpurun_out/r10b_ab.txt
cat gpurun_out/r10b_ab.txt
export UNET_SIDE_POLITE=83000 UNET_WZ_P11=100000000 UNET_PACK_GRID=256
bash profiles/profile_step_clean.sh r10b > /dev/null 2>&1
python3 profiles/timeline.py gpurun_out/prof_r10b_clean/runc_kernel_trace.csv 2 400 > gpurun_out/r10b_timeline.txt 2>&1
cd /tmp && export TMPDIR=/tmp
UNET_NO_SIDE_STREAM=1 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_r10b_solo -o runc -- python3 $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-profile --no-kernels --batch 0 > $R/gpurun_out/prof_r10b_solo.log 2>&1
cd $R
python3 profiles/stretch.py gpurun_out/prof_r10b_clean/runc_kernel_trace.csv gpurun_out/prof_r10b_solo/runc_kernel_trace.csv 8 60 > gpurun_out/r10b_stretch.txt 2>&1
head -3 gpurun_out/r10b_stretch.txt | cut -c1-250
