# PMC counters of the round-4 kernels, one GPU call:  gpurun --timeout 1200 -- 'bash profiles/collect_r18_counters.sh [tag]'
set -x
R=$GRAFT_REPO_ROOT
cd $R
T=${1:-r18}
# pipe + traffic counters of the new kernels at the 128^3 <-> 64^3 boundary (op-level launches: the dgrad without the accumulate / statistics tiles)
bash profiles/collect_counters.sh ${T}_s2_fwd_16to32_128 "k_s2_gather<3" profiles/op_kernel.py fwd 16 32 128 > /dev/null 2>&1
bash profiles/collect_counters.sh ${T}_s2_dgrad_16to32_128 "k_s2_scatter<0" profiles/op_kernel.py dgrad 16 32 128 > /dev/null 2>&1
bash profiles/collect_counters.sh ${T}_s2_wgrad_16to32_128 "k_s2_wgrad<3" profiles/op_kernel.py wgrad 16 32 128 > /dev/null 2>&1
bash profiles/collect_counters.sh ${T}_convt_fwd_32to16_64 "k_s2_scatter<1" profiles/op_kernel.py convt_fwd 32 16 64 > /dev/null 2>&1
bash profiles/collect_counters.sh ${T}_convt_dgrad_32to16_64 "k_s2_gather<2" profiles/op_kernel.py convt_dgrad 32 16 64 > /dev/null 2>&1
bash profiles/collect_counters.sh ${T}_convt_wgrad_32to16_64 "k_s2_wgrad<2" profiles/op_kernel.py convt_wgrad 32 16 64 > /dev/null 2>&1
UNET_OP_POLITE=1 bash profiles/collect_counters.sh ${T}_wgrad_zd_polite_32to16_128 "k_mfma_wgrad_zd" profiles/wgrad_kernel.py 32 16 128 > /dev/null 2>&1
ls gpurun_out/${T}_*counters.txt
