R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "micro or in_flight or sum_buffers or golden or convt_ops or conv3d_ops" > gpurun_out/r07g_tests.log 2>&1
tail -3 gpurun_out/r07g_tests.log
for cfg in "512 0" "256 0" "1024 0" "512 1" "1024 1" "2048 1" "512 2" "1024 2"; do
set -- $cfg
echo "== UNET_WGRAD_BLOCKS=$1 UNET_WGRAD_TS12=$2"
UNET_WGRAD_BLOCKS=$1 UNET_WGRAD_TS12=$2 python profiles/bench_wgrad.py 2>/dev/null | grep -E "s2|conv_trans" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('  %-48s %7.1f us  %6.0f GB/s'%(d['shape'], d['ms']*1e3, d['algorithmic_GBps']))"
done > gpurun_out/r07g_wgrad_sweep.txt 2>&1
cat gpurun_out/r07g_wgrad_sweep.txt
