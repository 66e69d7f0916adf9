# 24 + 24 alternating bench invocations (30 timed steps each) with / without the deep-level kernels: do multi-ms outliers follow the kernel set?
for i in $(seq 1 24); do
  for v in deep halo; do
    if [ $v = deep ]; then unset UNET_NO_DEEP_KERNELS; else export UNET_NO_DEEP_KERNELS=1; fi
    python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-cpp-host --no-profile --no-kernels --batch 0 2>/dev/null | python3 -c "import json,sys; print('$v', '%.3f' % json.loads(sys.stdin.read())['ms_per_step'])"
  done
done
